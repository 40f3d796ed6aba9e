// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// CPU restatement of the reference's interior-point solver:
//   include/sleipnir/optimization/solver/interior_point.hpp:63-878
//   .../solver/util/kkt_error.hpp:92-251
//   .../solver/util/filter.hpp:17-212
//   .../solver/util/fraction_to_the_boundary_rule.hpp:19-43
//   .../solver/util/is_locally_infeasible.hpp:17-60
//   .../solver/util/problem_scaling.hpp:21-115
//   .../solver/util/append_as_triplets.hpp:26-64
//   .../solver/util/feasibility_restoration.hpp:26-101,347-628
//   .../solver/util/lagrange_multiplier_estimate.hpp:56-133
// Pinned transitively by the reference's whole-solve known answers
// (tests/test_oracle_solves.py cites each one).
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <limits>
#include <tuple>
#include <vector>

#include "ldlt.hpp"
#include "sparse.hpp"

namespace orc {

// exit_status.hpp:13-43
enum class ExitStatus : int {
  SUCCESS = 0,
  CALLBACK_REQUESTED_STOP = 1,
  TOO_FEW_DOFS = -1,
  LOCALLY_INFEASIBLE = -2,
  GLOBALLY_INFEASIBLE = -3,
  FACTORIZATION_FAILED = -4,
  LINE_SEARCH_FAILED = -5,
  FEASIBILITY_RESTORATION_FAILED = -6,
  NONFINITE_INITIAL_GUESS = -7,
  DIVERGING_ITERATES = -8,
  MAX_ITERATIONS_EXCEEDED = -9,
  TIMEOUT = -10,
};

// options.hpp:13-38
struct Options {
  double tolerance = 1e-8;
  int max_iterations = 5000;
  double timeout = std::numeric_limits<double>::infinity();  // seconds
  bool feasible_ipm = false;
  bool diagnostics = false;
};

// problem_scaling.hpp
struct ProblemScaling {
  double f = 1.0;
  Vec c_e, c_i;
  ProblemScaling() = default;
  ProblemScaling(double f_, Vec ce, Vec ci) : f(f_), c_e(std::move(ce)), c_i(std::move(ci)) {}
  ProblemScaling(const Vec& g, const CSC& A_e, const CSC& A_i) {
    constexpr double g_max = 100.0;
    f = std::min(1.0, g_max / norm_inf(g));
    c_e = sparse_inf_norms(A_e);
    for (auto& v : c_e) v = std::min(g_max / v, 1.0);
    c_i = sparse_inf_norms(A_i);
    for (auto& v : c_i) v = std::min(g_max / v, 1.0);
  }
  bool is_identity() const { return f == 1.0 && c_e.empty() && c_i.empty(); }
};

// iteration_info.hpp:13-41
struct IterationInfo {
  int iteration;
  const Vec& x;
  const Vec& s;
  const Vec& y;
  const Vec& z;
  const Vec& g;
  const CSC& H;
  const CSC& A_e;
  const CSC& A_i;
};
using IterationCallback = std::function<bool(const IterationInfo&)>;

// interior_point_matrix_callbacks.hpp:18-250
struct MatrixCallbacks {
  int num_decision_variables = 0;
  int num_equality_constraints = 0;
  int num_inequality_constraints = 0;
  std::function<double(const Vec&)> f;
  std::function<Vec(const Vec&)> g;  // dense here; the reference's SparseVector is used densely
  std::function<CSC(const Vec&, const Vec&, const Vec&)> H;
  std::function<CSC(const Vec&, const Vec&, const Vec&)> H_c;
  std::function<Vec(const Vec&)> c_e;
  std::function<CSC(const Vec&)> A_e;
  std::function<Vec(const Vec&)> c_i;
  std::function<CSC(const Vec&)> A_i;
  ProblemScaling scaling;
};

// ----------------------------------------------------------------------------
// kkt_error.hpp
// ----------------------------------------------------------------------------
enum class KKTErrorType { INF_NORM_SCALED, ONE_NORM };

inline Vec axpy(const Vec& a, double alpha, const Vec& b) {  // a + alpha b
  Vec r(a.size());
  for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] + alpha * b[i];
  return r;
}
inline Vec vsub(const Vec& a, const Vec& b) { return axpy(a, -1.0, b); }
inline Vec cwise_mul(const Vec& a, const Vec& b) {
  Vec r(a.size());
  for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] * b[i];
  return r;
}

// kkt_error.hpp:92-146
template <KKTErrorType T>
double kkt_error(const Vec& g, const CSC& A_e, const Vec& c_e, const CSC& A_i, const Vec& c_i,
                 const Vec& s, const Vec& y, const Vec& z, double mu) {
  Vec dual = vsub(vsub(g, spmv_t(A_e, y)), spmv_t(A_i, z));
  Vec compl_(s.size());
  for (size_t i = 0; i < s.size(); ++i) compl_[i] = s[i] * z[i] - mu;
  Vec cis = vsub(c_i, s);
  if constexpr (T == KKTErrorType::INF_NORM_SCALED) {
    constexpr double s_max = 100.0;
    double s_d = std::max(s_max, (norm_1(y) + norm_1(z)) / double(y.size() + z.size())) / s_max;
    double s_c = std::max(s_max, norm_1(z) / double(z.size())) / s_max;
    return std::max({norm_inf(dual) / s_d, norm_inf(compl_) / s_c, norm_inf(c_e), norm_inf(cis)});
  } else {
    return norm_1(dual) + norm_1(compl_) + norm_1(c_e) + norm_1(cis);
  }
}

// kkt_error.hpp:216-251
template <KKTErrorType T>
double unscaled_kkt_error(const ProblemScaling& scaling, const Vec& g, const CSC& A_e,
                          const Vec& c_e, const CSC& A_i, const Vec& c_i, const Vec& s,
                          const Vec& y, const Vec& z, double mu) {
  if (scaling.is_identity()) return kkt_error<T>(g, A_e, c_e, A_i, c_i, s, y, z, mu);
  const double inv_d_f = 1.0 / scaling.f;
  Vec inv_d_c_e(scaling.c_e.size()), inv_d_c_i(scaling.c_i.size());
  for (size_t i = 0; i < inv_d_c_e.size(); ++i) inv_d_c_e[i] = 1.0 / scaling.c_e[i];
  for (size_t i = 0; i < inv_d_c_i.size(); ++i) inv_d_c_i[i] = 1.0 / scaling.c_i[i];
  Vec g_u(g.size());
  for (size_t i = 0; i < g.size(); ++i) g_u[i] = inv_d_f * g[i];
  CSC A_e_u = row_scaled(inv_d_c_e, A_e);
  Vec c_e_u = cwise_mul(inv_d_c_e, c_e);
  CSC A_i_u = row_scaled(inv_d_c_i, A_i);
  Vec c_i_u = cwise_mul(inv_d_c_i, c_i);
  Vec s_u = cwise_mul(inv_d_c_i, s);
  Vec y_u = cwise_mul(scaling.c_e, y);
  for (auto& v : y_u) v *= inv_d_f;
  Vec z_u = cwise_mul(scaling.c_i, z);
  for (auto& v : z_u) v *= inv_d_f;
  return kkt_error<T>(g_u, A_e_u, c_e_u, A_i_u, c_i_u, s_u, y_u, z_u, inv_d_f * mu);
}

// ----------------------------------------------------------------------------
// filter.hpp
// ----------------------------------------------------------------------------
struct FilterEntry {
  double cost = 0.0;
  double constraint_violation = 0.0;
  FilterEntry() = default;
  FilterEntry(double c, double v) : cost(c), constraint_violation(v) {}
  // filter.hpp:48-53
  FilterEntry(double f, const Vec& s, const Vec& c_e, const Vec& c_i, double mu) {
    double logsum = 0.0;
    for (double v : s) logsum += std::log(v);
    cost = f - mu * logsum;
    constraint_violation = norm_1(c_e) + norm_1(vsub(c_i, s));
  }
  bool dominated_by(const FilterEntry& e) const {
    return e.cost <= cost && e.constraint_violation <= constraint_violation;
  }
};

class Filter {
 public:
  double min_constraint_violation;
  double max_constraint_violation;
  explicit Filter(double initial_constraint_violation = 0.0) {
    min_constraint_violation = 1e-4 * std::max(1.0, initial_constraint_violation);
    max_constraint_violation = 1e4 * std::max(1.0, initial_constraint_violation);
  }
  void reset() {
    m_filter.clear();
    m_last_rejection_due_to_filter = false;
  }
  // filter.hpp:109-172
  bool try_add(const FilterEntry& current, const FilterEntry& trial, double D_phi, double alpha) {
    if (!std::isfinite(trial.cost) || trial.constraint_violation > max_constraint_violation)
      return false;
    constexpr double s_phi = 2.3, s_theta = 1.1;
    bool switching_condition =
        D_phi < 0.0 &&
        alpha * std::pow(-D_phi, s_phi) > std::pow(current.constraint_violation, s_theta);
    constexpr double eta_phi = 1e-8;
    bool armijo_condition = trial.cost <= current.cost + eta_phi * alpha * D_phi;
    double phi = std::pow(alpha, 1.5);
    bool sufficient_decrease =
        trial.cost <= current.cost - phi * gamma_cost * current.constraint_violation ||
        trial.constraint_violation <= (1.0 - phi * gamma_constraint) * current.constraint_violation;
    if (current.constraint_violation <= min_constraint_violation && switching_condition) {
      if (!armijo_condition) {
        m_last_rejection_due_to_filter = false;
        return false;
      }
    } else if (!sufficient_decrease) {
      m_last_rejection_due_to_filter = false;
      return false;
    }
    if (in_filter(trial)) {
      m_last_rejection_due_to_filter = true;
      return false;
    }
    if (!switching_condition || !armijo_condition) {
      add(FilterEntry{current.cost - phi * gamma_cost * current.constraint_violation,
                      (1.0 - phi * gamma_constraint) * current.constraint_violation});
    }
    return true;
  }
  bool last_rejection_due_to_filter() const { return m_last_rejection_due_to_filter; }

 private:
  static constexpr double gamma_cost = 1e-8;
  static constexpr double gamma_constraint = 1e-5;
  std::vector<FilterEntry> m_filter;
  bool m_last_rejection_due_to_filter = false;
  void add(const FilterEntry& entry) {
    m_filter.erase(std::remove_if(m_filter.begin(), m_filter.end(),
                                  [&](const FilterEntry& e) { return e.dominated_by(entry); }),
                   m_filter.end());
    m_filter.push_back(entry);
  }
  bool in_filter(const FilterEntry& entry) const {
    return std::any_of(m_filter.begin(), m_filter.end(),
                       [&](const FilterEntry& e) { return entry.dominated_by(e); });
  }
};

// fraction_to_the_boundary_rule.hpp:19-43
inline double fraction_to_the_boundary_rule(const Vec& x, const Vec& p, double tau) {
  double alpha = 1.0;
  for (size_t i = 0; i < x.size(); ++i) {
    if (alpha * p[i] < -tau * x[i]) alpha = -tau / p[i] * x[i];
  }
  return alpha;
}

// is_locally_infeasible.hpp:17-60
inline bool is_equality_locally_infeasible(const CSC& A_e, const Vec& c_e) {
  return A_e.rows > 0 && norm_2(spmv_t(A_e, c_e)) < 1e-6 && norm_2(c_e) > 1e-2;
}
inline bool is_inequality_locally_infeasible(const CSC& A_i, const Vec& c_i) {
  if (A_i.rows > 0) {
    Vec c_plus(c_i.size());
    for (size_t i = 0; i < c_i.size(); ++i) c_plus[i] = std::min(c_i[i], 0.0);
    if (norm_2(spmv_t(A_i, c_plus)) < 1e-6 && norm_2(c_plus) > 1e-6) return true;
  }
  return false;
}

// append_as_triplets.hpp:26-48: vertically stacked blocks, column-interleaved
inline void append_as_triplets(std::vector<Triplet>& triplets, int row_offset, int col_offset,
                               std::initializer_list<const CSC*> mats) {
  std::vector<int> offs;
  int off = 0;
  for (const CSC* m : mats) {
    offs.push_back(off);
    off += m->rows;
  }
  const CSC* first = *mats.begin();
  for (int col = 0; col < first->cols; ++col) {
    size_t i = 0;
    for (const CSC* m : mats) {
      for (int p = m->colptr[col]; p < m->colptr[col + 1]; ++p)
        triplets.push_back({row_offset + offs[i] + m->rowidx[p], col_offset + col, m->val[p]});
      ++i;
    }
  }
}
inline void append_diagonal_as_triplets(std::vector<Triplet>& triplets, int row_offset,
                                        int col_offset, const Vec& diag) {
  for (int r = 0; r < static_cast<int>(diag.size()); ++r)
    triplets.push_back({row_offset + r, col_offset + r, diag[r]});
}

// The pieces of one Newton step, exposed so parity tests can compare the product's
// kernels against them one by one (interior_point.hpp:426-482).
struct NewtonStepPieces {
  CSC lhs;
  Vec rhs;
  Vec p_x, p_y, p_s, p_z;
  double delta = 0.0, gamma = 0.0;
  int factorizations = 0;
  Info info = Success;
};

// interior_point.hpp:426-440
inline CSC build_kkt_lhs(const CSC& H, const CSC& A_e, const CSC& A_i, const Vec& s, const Vec& z) {
  Vec sigma(s.size());
  for (size_t i = 0; i < s.size(); ++i) sigma[i] = (1.0 / s[i]) * z[i];
  // A_iᵀ Σ A_i, lower triangle
  CSC AiT = transpose(A_i);
  CSC prod = lower_triangle(multiply(multiply(AiT, diag_matrix(sigma)), A_i));
  CSC top_left = add(H, prod);
  std::vector<Triplet> triplets;
  triplets.reserve(top_left.nnz() + A_e.nnz());
  append_as_triplets(triplets, 0, 0, {&top_left, &A_e});
  int dim = H.rows + A_e.rows;
  return from_triplets(dim, dim, triplets);
}

// interior_point.hpp:444-448
inline Vec build_kkt_rhs(const Vec& g, const CSC& A_e, const CSC& A_i, const Vec& c_e,
                         const Vec& c_i, const Vec& s, const Vec& y, const Vec& z, double mu) {
  int n = static_cast<int>(g.size());
  Vec t(s.size());
  for (size_t i = 0; i < s.size(); ++i) {
    double sigma = (1.0 / s[i]) * z[i];
    t[i] = -sigma * c_i[i] + mu * (1.0 / s[i]) + z[i];
  }
  Vec Aey = spmv_t(A_e, y);
  Vec Ait = spmv_t(A_i, t);
  Vec rhs(n + y.size());
  for (int i = 0; i < n; ++i) rhs[i] = -g[i] + Aey[i] + Ait[i];
  for (size_t i = 0; i < y.size(); ++i) rhs[n + i] = -c_e[i];
  return rhs;
}

// interior_point.hpp:470-481
inline void back_substitute(const Vec& p, const CSC& A_i, const Vec& c_i_minus_s, const Vec& s,
                            const Vec& z, double mu, int n, int m_e, Vec& p_x, Vec& p_y,
                            Vec& p_s, Vec& p_z) {
  p_x.assign(p.begin(), p.begin() + n);
  p_y.resize(m_e);
  for (int i = 0; i < m_e; ++i) p_y[i] = -p[n + i];
  Vec Aipx = spmv(A_i, p_x);
  p_s.resize(s.size());
  p_z.resize(s.size());
  for (size_t i = 0; i < s.size(); ++i) {
    p_s[i] = c_i_minus_s[i] + Aipx[i];
    double sigma = (1.0 / s[i]) * z[i];
    p_z[i] = mu * (1.0 / s[i]) - z[i] - sigma * p_s[i];
  }
}

struct SolveStats {
  int iterations = 0;
  int factorizations = 0;
  int solves = 0;
  double t_ad = 0, t_build = 0, t_decomp = 0, t_solve = 0, t_linesearch = 0, t_total = 0;
};

ExitStatus feasibility_restoration(const MatrixCallbacks& matrices,
                                   std::vector<IterationCallback>& iteration_callbacks,
                                   const Options& options, Vec& x, Vec& s, Vec& y, Vec& z,
                                   double mu, int& iterations, SolveStats* stats);

// interior_point.hpp:123-863 (worker overload)
inline ExitStatus interior_point(const MatrixCallbacks& matrices,
                                 std::vector<IterationCallback>& iteration_callbacks,
                                 const Options& options, bool in_feasibility_restoration, Vec& x,
                                 Vec& s, Vec& y, Vec& z, double& mu, int& iterations,
                                 SolveStats* stats = nullptr,
                                 const std::vector<int>* user_perm = nullptr) {
  using clock = std::chrono::steady_clock;
  auto secs = [](clock::time_point a, clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  const auto solve_start_time = clock::now();
  const int n = matrices.num_decision_variables;
  const int m_e = matrices.num_equality_constraints;

  double f = matrices.f(x);
  Vec g = matrices.g(x);
  CSC H = matrices.H(x, y, z);
  Vec c_e = matrices.c_e(x);
  CSC A_e = matrices.A_e(x);
  Vec c_i = matrices.c_i(x);
  CSC A_i = matrices.A_i(x);

  Vec trial_x, trial_s, trial_y, trial_z;
  double trial_f;
  Vec trial_c_e, trial_c_i;

  if (m_e > n) return ExitStatus::TOO_FEW_DOFS;  // :274

  if (!std::isfinite(f) || !all_finite(g) || !all_finite(H) || !all_finite(c_e) ||
      !all_finite(A_e) || !all_finite(c_i) || !all_finite(A_i)) {
    return ExitStatus::NONFINITE_INITIAL_GUESS;  // :283-286
  }

  const double mu_min = matrices.scaling.f * options.tolerance / 10.0;  // :294
  constexpr double tau_min = 0.99;
  double tau = tau_min;

  Filter filter{norm_1(c_e) + norm_1(vsub(c_i, s))};  // :303

  auto update_barrier_parameter_and_reset_filter = [&] {  // :308-333
    constexpr double kappa_mu = 0.2;
    constexpr double theta_mu = 1.5;
    mu = std::max(mu_min, std::min(kappa_mu * mu, std::pow(mu, theta_mu)));
    tau = std::max(tau_min, 1.0 - mu);
    filter.reset();
  };

  const int lhs_rows = n + m_e;
  // :340-352
  int AiTAi_lower_nnz = lower_triangle(multiply(transpose(A_i), A_i)).nnz();
  RegularizedLDLT solver{
      double(H.nnz() + AiTAi_lower_nnz + A_e.nnz()) < 0.25 * double(lhs_rows) * double(lhs_rows),
      n, m_e, in_feasibility_restoration ? 0.0 : 1e-10};
  if (user_perm != nullptr && !user_perm->empty()) solver.set_permutation(*user_perm);

  constexpr double alpha_reduction_factor = 0.5;
  constexpr double alpha_min = 1e-7;
  int full_step_rejected_counter = 0;

  double E_0 = unscaled_kkt_error<KKTErrorType::INF_NORM_SCALED>(matrices.scaling, g, A_e, c_e,
                                                                  A_i, c_i, s, y, z, 0.0);

  while (E_0 > options.tolerance) {
    if (is_equality_locally_infeasible(A_e, c_e)) return ExitStatus::LOCALLY_INFEASIBLE;
    if (is_inequality_locally_infeasible(A_i, c_i)) return ExitStatus::LOCALLY_INFEASIBLE;
    if (norm_inf(x) > 1e10 || !all_finite(x) || norm_inf(s) > 1e10 || !all_finite(s))
      return ExitStatus::DIVERGING_ITERATES;

    for (const auto& callback : iteration_callbacks) {
      if (callback({iterations, x, s, y, z, g, H, A_e, A_i}))
        return ExitStatus::CALLBACK_REQUESTED_STOP;
    }

    auto t0 = clock::now();
    CSC lhs = build_kkt_lhs(H, A_e, A_i, s, z);
    Vec rhs = build_kkt_rhs(g, A_e, A_i, c_e, c_i, s, y, z, mu);
    Vec sigma(s.size());
    for (size_t i = 0; i < s.size(); ++i) sigma[i] = (1.0 / s[i]) * z[i];
    auto t1 = clock::now();

    Vec p_x, p_s, p_y, p_z;
    double alpha_max = 1.0, alpha = 1.0, alpha_z = 1.0;
    bool call_feasibility_restoration = false;

    if (solver.compute(lhs).info() != Success) return ExitStatus::FACTORIZATION_FAILED;  // :463
    auto t2 = clock::now();
    if (stats) {
      stats->t_build += secs(t0, t1);
      stats->t_decomp += secs(t1, t2);
      stats->factorizations += solver.factorizations();
    }

    auto compute_step = [&](Vec& px, Vec& ps, Vec& py, Vec& pz, const Vec& c_i_minus_s) {
      Vec p = solver.solve(rhs);
      if (stats) ++stats->solves;
      back_substitute(p, A_i, c_i_minus_s, s, z, mu, n, m_e, px, py, ps, pz);
    };
    compute_step(p_x, p_s, p_y, p_z, vsub(c_i, s));
    auto t3 = clock::now();
    if (stats) stats->t_solve += secs(t2, t3);

    alpha_max = fraction_to_the_boundary_rule(s, p_s, tau);  // :488
    alpha = alpha_max;
    if (alpha < alpha_min) call_feasibility_restoration = true;
    alpha_z = fraction_to_the_boundary_rule(z, p_z, tau);  // :497

    const FilterEntry current_entry{f, s, c_e, c_i, mu};

    // :508-509
    double sinv_dot_ps = 0.0;
    for (size_t i = 0; i < s.size(); ++i) sinv_dot_ps += (1.0 / s[i]) * p_s[i];
    const double D_phi = dot(g, p_x) - mu * sinv_dot_ps;

    while (true) {  // :512
      trial_x = axpy(x, alpha, p_x);
      trial_c_i = matrices.c_i(trial_x);
      bool all_pos = true;
      for (double v : c_i) all_pos = all_pos && (v > 0.0);
      if (options.feasible_ipm && all_pos) {
        trial_s = trial_c_i;
      } else {
        trial_s = axpy(s, alpha, p_s);
      }
      trial_y = axpy(y, alpha_z, p_y);
      trial_z = axpy(z, alpha_z, p_z);

      trial_f = matrices.f(trial_x);
      trial_c_e = matrices.c_e(trial_x);

      if (!std::isfinite(trial_f) || !all_finite(trial_c_e) || !all_finite(trial_c_i)) {
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) {
          call_feasibility_restoration = true;
          break;
        }
        continue;
      }

      FilterEntry trial_entry{trial_f, trial_s, trial_c_e, trial_c_i, mu};
      if (filter.try_add(current_entry, trial_entry, D_phi, alpha)) break;

      double prev_constraint_violation = norm_1(c_e) + norm_1(vsub(c_i, s));
      double next_constraint_violation = norm_1(trial_c_e) + norm_1(vsub(trial_c_i, trial_s));

      // Second-order corrections :566-668
      if (alpha == alpha_max && next_constraint_violation >= prev_constraint_violation) {
        Vec soc_px = p_x, soc_ps = p_s, soc_py = p_y, soc_pz = p_z;
        double alpha_soc = alpha;
        double alpha_z_soc = alpha_z;
        Vec c_e_soc = c_e;
        Vec c_i_minus_s_soc = vsub(c_i, s);
        double soc_constraint_violation = next_constraint_violation;
        bool step_acceptable = false;
        for (int soc_iteration = 0; soc_iteration < 5 && !step_acceptable; ++soc_iteration) {
          for (size_t i = 0; i < c_e_soc.size(); ++i)
            c_e_soc[i] = alpha_soc * c_e_soc[i] + trial_c_e[i];
          for (size_t i = 0; i < c_i_minus_s_soc.size(); ++i)
            c_i_minus_s_soc[i] = alpha_soc * c_i_minus_s_soc[i] + trial_c_i[i] - trial_s[i];
          {
            Vec t(s.size());
            for (size_t i = 0; i < s.size(); ++i)
              t[i] = mu * (1.0 / s[i]) - sigma[i] * c_i_minus_s_soc[i];
            Vec Aey = spmv_t(A_e, y);
            Vec Ait = spmv_t(A_i, t);
            for (int i = 0; i < n; ++i) rhs[i] = -g[i] + Aey[i] + Ait[i];
            for (int i = 0; i < m_e; ++i) rhs[n + i] = -c_e_soc[i];
          }
          compute_step(soc_px, soc_ps, soc_py, soc_pz, c_i_minus_s_soc);

          alpha_soc = fraction_to_the_boundary_rule(s, soc_ps, tau);
          alpha_z_soc = fraction_to_the_boundary_rule(z, soc_pz, tau);

          trial_x = axpy(x, alpha_soc, soc_px);
          trial_s = axpy(s, alpha_soc, soc_ps);
          trial_y = axpy(y, alpha_z_soc, soc_py);
          trial_z = axpy(z, alpha_z_soc, soc_pz);

          trial_f = matrices.f(trial_x);
          trial_c_e = matrices.c_e(trial_x);
          trial_c_i = matrices.c_i(trial_x);

          FilterEntry soc_trial{trial_f, trial_s, trial_c_e, trial_c_i, mu};
          if (filter.try_add(current_entry, soc_trial, D_phi, alpha)) {
            p_x = soc_px;
            p_s = soc_ps;
            p_y = soc_py;
            p_z = soc_pz;
            alpha = alpha_soc;
            alpha_z = alpha_z_soc;
            step_acceptable = true;
            break;
          }
          constexpr double kappa_soc = 0.99;
          next_constraint_violation = norm_1(trial_c_e) + norm_1(vsub(trial_c_i, trial_s));
          if (next_constraint_violation > kappa_soc * soc_constraint_violation) break;
          soc_constraint_violation = next_constraint_violation;
        }
        if (step_acceptable) break;
      }

      if (alpha == alpha_max) ++full_step_rejected_counter;

      // :677-684
      if (full_step_rejected_counter >= 4 &&
          filter.max_constraint_violation > current_entry.constraint_violation / 10.0 &&
          filter.last_rejection_due_to_filter()) {
        filter.max_constraint_violation *= 0.1;
        filter.reset();
        continue;
      }

      alpha *= alpha_reduction_factor;

      // :691-716
      if (alpha < alpha_min) {
        double current_kkt_error =
            kkt_error<KKTErrorType::ONE_NORM>(g, A_e, c_e, A_i, c_i, s, y, z, mu);
        trial_x = axpy(x, alpha_max, p_x);
        trial_s = axpy(s, alpha_max, p_s);
        trial_y = axpy(y, alpha_z, p_y);
        trial_z = axpy(z, alpha_z, p_z);
        trial_f = matrices.f(trial_x);
        trial_c_e = matrices.c_e(trial_x);
        trial_c_i = matrices.c_i(trial_x);
        double next_kkt_error = kkt_error<KKTErrorType::ONE_NORM>(
            matrices.g(trial_x), matrices.A_e(trial_x), trial_c_e, matrices.A_i(trial_x),
            trial_c_i, trial_s, trial_y, trial_z, mu);
        if (next_kkt_error <= 0.999 * current_kkt_error) break;
        call_feasibility_restoration = true;
        break;
      }
    }
    auto t4 = clock::now();
    if (stats) stats->t_linesearch += secs(t3, t4);

    if (call_feasibility_restoration) {  // :721-771
      if (in_feasibility_restoration) return ExitStatus::FEASIBILITY_RESTORATION_FAILED;

      FilterEntry initial_entry{matrices.f(x), s, c_e, c_i, mu};
      std::vector<IterationCallback> callbacks;
      for (auto& cb : iteration_callbacks) callbacks.push_back(cb);
      callbacks.push_back([&](const IterationInfo& info) {
        Vec tx(info.x.begin(), info.x.begin() + n);
        Vec ts(info.s.begin(), info.s.begin() + matrices.num_inequality_constraints);
        Vec tce = matrices.c_e(tx);
        Vec tci = matrices.c_i(tx);
        FilterEntry trial_entry{matrices.f(tx), ts, tce, tci, mu};
        double sinv_dot = 0.0;
        for (size_t i = 0; i < s.size(); ++i) sinv_dot += (1.0 / s[i]) * (ts[i] - s[i]);
        const double D_phi_restoration = dot(g, vsub(tx, x)) - mu * sinv_dot;
        return trial_entry.constraint_violation < 0.9 * initial_entry.constraint_violation &&
               filter.try_add(initial_entry, trial_entry, D_phi_restoration, alpha);
      });
      auto status =
          feasibility_restoration(matrices, callbacks, options, x, s, y, z, mu, iterations, stats);
      if (status != ExitStatus::SUCCESS) return status;
      f = matrices.f(x);
      c_e = matrices.c_e(x);
      c_i = matrices.c_i(x);
    } else {
      if (alpha == alpha_max) full_step_rejected_counter = 0;
      x = trial_x;
      s = trial_s;
      y = trial_y;
      z = trial_z;
      // :797-801
      for (size_t row = 0; row < z.size(); ++row) {
        constexpr double kappa_sigma = 1e10;
        z[row] = std::clamp(z[row], 1.0 / kappa_sigma * mu / s[row], kappa_sigma * mu / s[row]);
      }
      f = trial_f;
      c_e = trial_c_e;
      c_i = trial_c_i;
    }

    // :809-812 AD refresh
    auto t5 = clock::now();
    A_e = matrices.A_e(x);
    A_i = matrices.A_i(x);
    g = matrices.g(x);
    H = matrices.H(x, y, z);
    auto t6 = clock::now();
    if (stats) stats->t_ad += secs(t5, t6);

    E_0 = unscaled_kkt_error<KKTErrorType::INF_NORM_SCALED>(matrices.scaling, g, A_e, c_e, A_i,
                                                            c_i, s, y, z, 0.0);
    if (E_0 > options.tolerance) {  // :819-832
      constexpr double kappa_eps = 10.0;
      double E_mu = kkt_error<KKTErrorType::INF_NORM_SCALED>(g, A_e, c_e, A_i, c_i, s, y, z, mu);
      while (mu > mu_min && E_mu <= kappa_eps * mu) {
        update_barrier_parameter_and_reset_filter();
        E_mu = kkt_error<KKTErrorType::INF_NORM_SCALED>(g, A_e, c_e, A_i, c_i, s, y, z, mu);
      }
    }

    ++iterations;
    if (stats) stats->iterations = iterations;
    if (iterations >= options.max_iterations) return ExitStatus::MAX_ITERATIONS_EXCEEDED;
    if (secs(solve_start_time, clock::now()) > options.timeout) return ExitStatus::TIMEOUT;
  }
  return ExitStatus::SUCCESS;
}

// interior_point.hpp:63-87 (entry overload)
inline ExitStatus interior_point(const MatrixCallbacks& matrices,
                                 std::vector<IterationCallback>& iteration_callbacks,
                                 const Options& options, Vec& x, SolveStats* stats = nullptr,
                                 const std::vector<int>* user_perm = nullptr,
                                 Vec* s_out = nullptr, Vec* y_out = nullptr, Vec* z_out = nullptr) {
  Vec s(matrices.num_inequality_constraints, 1.0);
  Vec y(matrices.num_equality_constraints, 0.0);
  Vec z(matrices.num_inequality_constraints, 1.0);
  double mu = 0.1 * matrices.scaling.f;
  int iterations = 0;
  auto st = interior_point(matrices, iteration_callbacks, options, false, x, s, y, z, mu,
                           iterations, stats, user_perm);
  if (s_out) *s_out = s;
  if (y_out) *y_out = y;
  if (z_out) *z_out = z;
  return st;
}

// ----------------------------------------------------------------------------
// sqp.hpp:98-604 — problems with equality constraints only (problem.hpp:403).  The
// MatrixCallbacks of the interior-point method are used with no inequality rows.
// ----------------------------------------------------------------------------
inline ExitStatus sqp(const MatrixCallbacks& matrices, std::vector<IterationCallback>& iteration_callbacks,
                      const Options& options, Vec& x, Vec& y, int& iterations, SolveStats* stats = nullptr,
                      const std::vector<int>* user_perm = nullptr) {
  using clock = std::chrono::steady_clock;
  const auto solve_start_time = clock::now();
  const int n = matrices.num_decision_variables;
  const int m_e = matrices.num_equality_constraints;
  const Vec none;
  const CSC no_rows(0, n);

  double f = matrices.f(x);
  Vec g = matrices.g(x);
  CSC H = matrices.H(x, y, none);
  Vec c_e = matrices.c_e(x);
  CSC A_e = matrices.A_e(x);
  Vec trial_x, trial_y, trial_c_e;
  double trial_f;

  if (m_e > n) return ExitStatus::TOO_FEW_DOFS;  // :205-210
  if (!std::isfinite(f) || !all_finite(g) || !all_finite(H) || !all_finite(c_e) || !all_finite(A_e))
    return ExitStatus::NONFINITE_INITIAL_GUESS;  // :213-216

  Filter filter{norm_1(c_e)};  // :220
  const int lhs_rows = n + m_e;
  RegularizedLDLT solver{double(H.nnz() + A_e.nnz()) < 0.25 * double(lhs_rows) * double(lhs_rows), n, m_e};  // :238-240
  if (user_perm != nullptr && !user_perm->empty()) solver.set_permutation(*user_perm);
  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-7;
  int full_step_rejected_counter = 0;
  auto error = [&](const Vec& gg, const CSC& Ae, const Vec& ce, const Vec& yy) {  // kkt_error.hpp (SQP overloads)
    return unscaled_kkt_error<KKTErrorType::INF_NORM_SCALED>(matrices.scaling, gg, Ae, ce, no_rows, none, none, yy,
                                                             none, 0.0);
  };
  double E_0 = error(g, A_e, c_e, y);

  while (E_0 > options.tolerance) {
    if (is_equality_locally_infeasible(A_e, c_e)) return ExitStatus::LOCALLY_INFEASIBLE;  // :277
    if (norm_inf(x) > 1e10 || !all_finite(x)) return ExitStatus::DIVERGING_ITERATES;      // :288
    for (const auto& callback : iteration_callbacks)
      if (callback({iterations, x, none, y, none, g, H, A_e, no_rows})) return ExitStatus::CALLBACK_REQUESTED_STOP;

    // :305-325  lhs = [H A_e^T; A_e 0] (lower), rhs = -[g - A_e^T y; c_e]
    CSC lhs = build_kkt_lhs(H, A_e, no_rows, none, none);
    Vec rhs = build_kkt_rhs(g, A_e, no_rows, c_e, none, none, y, none, 0.0);
    if (solver.compute(lhs).info() != Success) return ExitStatus::FACTORIZATION_FAILED;  // :336
    if (stats) stats->factorizations += solver.factorizations();
    auto compute_step = [&](Vec& px, Vec& py) {
      Vec p = solver.solve(rhs);
      if (stats) ++stats->solves;
      px.assign(p.begin(), p.begin() + n);
      py.resize(m_e);
      for (int j = 0; j < m_e; ++j) py[j] = -p[n + j];
    };
    Vec p_x, p_y;
    compute_step(p_x, p_y);

    constexpr double alpha_max = 1.0;
    double alpha = alpha_max;
    bool call_feasibility_restoration = false;
    const FilterEntry current_entry{f, norm_1(c_e)};
    const double D_phi = dot(g, p_x);  // :360

    while (true) {
      trial_x = axpy(x, alpha, p_x);
      trial_y = axpy(y, alpha, p_y);
      trial_f = matrices.f(trial_x);
      trial_c_e = matrices.c_e(trial_x);
      if (!std::isfinite(trial_f) || !all_finite(trial_c_e)) {  // :373-384
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) {
          call_feasibility_restoration = true;
          break;
        }
        continue;
      }
      if (filter.try_add(current_entry, FilterEntry{trial_f, norm_1(trial_c_e)}, D_phi, alpha)) break;

      const double prev_violation = norm_1(c_e);
      double next_violation = norm_1(trial_c_e);
      if (alpha == alpha_max && next_violation >= prev_violation) {  // :397-468 second-order corrections
        Vec soc_px = p_x, soc_py = p_y, c_e_soc = c_e;
        const double alpha_soc = alpha;
        double soc_violation = next_violation;
        bool step_acceptable = false;
        for (int soc_iteration = 0; soc_iteration < 5 && !step_acceptable; ++soc_iteration) {
          for (int j = 0; j < m_e; ++j) c_e_soc[j] = alpha_soc * c_e_soc[j] + trial_c_e[j];
          for (int j = 0; j < m_e; ++j) rhs[n + j] = -c_e_soc[j];
          compute_step(soc_px, soc_py);
          trial_x = axpy(x, alpha_soc, soc_px);
          trial_y = axpy(y, alpha_soc, soc_py);
          trial_f = matrices.f(trial_x);
          trial_c_e = matrices.c_e(trial_x);
          if (filter.try_add(current_entry, FilterEntry{trial_f, norm_1(trial_c_e)}, D_phi, alpha)) {
            p_x = soc_px;
            p_y = soc_py;
            alpha = alpha_soc;
            step_acceptable = true;
            break;
          }
          constexpr double kappa_soc = 0.99;
          next_violation = norm_1(trial_c_e);
          if (next_violation > kappa_soc * soc_violation) break;
          soc_violation = next_violation;
        }
        if (step_acceptable) break;
      }
      if (alpha == alpha_max) ++full_step_rejected_counter;
      if (full_step_rejected_counter >= 4 && filter.max_constraint_violation > current_entry.constraint_violation / 10.0 &&
          filter.last_rejection_due_to_filter()) {
        filter.max_constraint_violation *= 0.1;
        filter.reset();
        continue;
      }
      alpha *= alpha_reduction_factor;
      if (alpha < alpha_min) {  // :492-517
        const double current_kkt = kkt_error<KKTErrorType::ONE_NORM>(g, A_e, c_e, no_rows, none, none, y, none, 0.0);
        trial_x = axpy(x, alpha_max, p_x);
        trial_y = axpy(y, alpha_max, p_y);
        trial_f = matrices.f(trial_x);
        trial_c_e = matrices.c_e(trial_x);
        const double next_kkt = kkt_error<KKTErrorType::ONE_NORM>(matrices.g(trial_x), matrices.A_e(trial_x), trial_c_e,
                                                                  no_rows, none, none, trial_y, none, 0.0);
        if (next_kkt <= 0.999 * current_kkt) break;
        call_feasibility_restoration = true;
        break;
      }
    }

    if (call_feasibility_restoration) {  // :521-556
      const FilterEntry initial_entry{matrices.f(x), norm_1(c_e)};
      std::vector<IterationCallback> callbacks;
      for (auto& cb : iteration_callbacks) callbacks.push_back(cb);
      callbacks.push_back([&](const IterationInfo& info) {
        Vec tx(info.x.begin(), info.x.begin() + n);
        const Vec tce = matrices.c_e(tx);
        const FilterEntry trial_entry{matrices.f(tx), norm_1(tce)};
        const double D_phi_restoration = dot(g, vsub(tx, x));
        return trial_entry.constraint_violation < 0.9 * initial_entry.constraint_violation &&
               filter.try_add(initial_entry, trial_entry, D_phi_restoration, alpha);
      });
      // feasibility_restoration.hpp:103-345: the interior-point variant with no inequality rows of
      // the original problem and mu = tolerance / 10
      Vec s_none, z_none;
      const ExitStatus status = feasibility_restoration(matrices, callbacks, options, x, s_none, y, z_none,
                                                        options.tolerance / 10.0, iterations, stats);
      if (status != ExitStatus::SUCCESS) return status;
      f = matrices.f(x);
      c_e = matrices.c_e(x);
    } else {
      if (alpha == alpha_max) full_step_rejected_counter = 0;
      x = trial_x;
      y = trial_y;
      f = trial_f;
      c_e = trial_c_e;
    }
    A_e = matrices.A_e(x);  // :574-577
    g = matrices.g(x);
    H = matrices.H(x, y, none);
    E_0 = error(g, A_e, c_e, y);
    ++iterations;
    if (iterations >= options.max_iterations) return ExitStatus::MAX_ITERATIONS_EXCEEDED;
    if (std::chrono::duration<double>(clock::now() - solve_start_time).count() > options.timeout) return ExitStatus::TIMEOUT;
  }
  return ExitStatus::SUCCESS;
}

// ----------------------------------------------------------------------------
// newton.hpp:51-292 — unconstrained problems (problem.hpp:335)
// ----------------------------------------------------------------------------
inline ExitStatus newton(const MatrixCallbacks& matrices, std::vector<IterationCallback>& iteration_callbacks,
                         const Options& options, Vec& x, int& iterations, SolveStats* stats = nullptr) {
  using clock = std::chrono::steady_clock;
  const auto solve_start_time = clock::now();
  const int n = matrices.num_decision_variables;
  const Vec none;
  const CSC no_rows(0, n);
  double f = matrices.f(x);
  Vec g = matrices.g(x);
  CSC H = matrices.H(x, none, none);
  if (!std::isfinite(f) || !all_finite(g) || !all_finite(H)) return ExitStatus::NONFINITE_INITIAL_GUESS;  // :125
  Filter filter{0.0};  // :131
  RegularizedLDLT solver{double(H.nnz()) < 0.25 * double(n) * double(n), n, 0};  // :133-135
  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-20;
  auto error = [&](const Vec& gg) {
    return unscaled_kkt_error<KKTErrorType::INF_NORM_SCALED>(matrices.scaling, gg, no_rows, none, no_rows, none, none,
                                                             none, none, 0.0);
  };
  double E_0 = error(g);
  Vec trial_x;
  double trial_f;
  while (E_0 > options.tolerance) {
    if (norm_inf(x) > 1e10 || !all_finite(x)) return ExitStatus::DIVERGING_ITERATES;
    for (const auto& callback : iteration_callbacks)
      if (callback({iterations, x, none, none, none, g, H, no_rows, no_rows})) return ExitStatus::CALLBACK_REQUESTED_STOP;
    if (solver.compute(H).info() != Success) return ExitStatus::FACTORIZATION_FAILED;
    if (stats) stats->factorizations += solver.factorizations();
    Vec neg_g(n);
    for (int i = 0; i < n; ++i) neg_g[i] = -g[i];
    Vec p_x = solver.solve(neg_g);  // :190
    if (stats) ++stats->solves;
    constexpr double alpha_max = 1.0;
    double alpha = alpha_max;
    const double D_phi = dot(g, p_x);
    while (true) {  // :201-243
      trial_x = axpy(x, alpha, p_x);
      trial_f = matrices.f(trial_x);
      if (!std::isfinite(trial_f)) {
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) return ExitStatus::LINE_SEARCH_FAILED;
        continue;
      }
      if (filter.try_add(FilterEntry{f, 0.0}, FilterEntry{trial_f, 0.0}, D_phi, alpha)) break;
      alpha *= alpha_reduction_factor;
      if (alpha < alpha_min) {
        const double current_kkt = norm_1(g);
        trial_x = axpy(x, alpha_max, p_x);
        const double next_kkt = norm_1(matrices.g(trial_x));
        if (next_kkt <= 0.999 * current_kkt) {
          trial_f = matrices.f(trial_x);
          break;
        }
        return ExitStatus::LINE_SEARCH_FAILED;
      }
    }
    x = trial_x;
    f = trial_f;
    g = matrices.g(x);
    H = matrices.H(x, none, none);
    E_0 = error(g);
    ++iterations;
    if (iterations >= options.max_iterations) return ExitStatus::MAX_ITERATIONS_EXCEEDED;
    if (std::chrono::duration<double>(clock::now() - solve_start_time).count() > options.timeout) return ExitStatus::TIMEOUT;
  }
  return ExitStatus::SUCCESS;
}

// ----------------------------------------------------------------------------
// lagrange_multiplier_estimate.hpp:56-133
// ----------------------------------------------------------------------------
inline std::pair<Vec, Vec> lagrange_multiplier_estimate(const Vec& g, const CSC& A_e,
                                                        const CSC& A_i, const Vec& s, double mu) {
  std::vector<Triplet> triplets;
  append_as_triplets(triplets, 0, 0, {&A_e, &A_i});
  Vec neg_s(s.size());
  for (size_t i = 0; i < s.size(); ++i) neg_s[i] = -s[i];
  append_diagonal_as_triplets(triplets, A_e.rows, A_i.cols, neg_s);
  CSC A_hat = from_triplets(A_e.rows + A_i.rows, A_e.cols + static_cast<int>(s.size()), triplets);
  CSC lhs = multiply(A_hat, transpose(A_hat));
  Vec rhs_temp(g.size() + s.size());
  for (size_t i = 0; i < g.size(); ++i) rhs_temp[i] = g[i];
  for (size_t i = 0; i < s.size(); ++i) rhs_temp[g.size() + i] = -mu;
  Vec rhs = spmv(A_hat, rhs_temp);
  SimplicialLDLT est;
  CSC lower = lower_triangle(lhs);
  est.analyze_pattern(lower);
  est.factorize(lower);
  Vec sol = est.solve(rhs);
  Vec y(sol.begin(), sol.begin() + A_e.rows);
  Vec z(sol.begin() + A_e.rows, sol.begin() + A_e.rows + s.size());
  for (size_t row = 0; row < z.size(); ++row) {
    constexpr double kappa_sigma = 1e10;
    z[row] = std::clamp(z[row], 1.0 / kappa_sigma * mu / s[row], kappa_sigma * mu / s[row]);
  }
  return {y, z};
}

// feasibility_restoration.hpp:26-101
inline std::pair<Vec, Vec> compute_p_n(const Vec& c, double rho, double mu) {
  Vec p(c.size()), n(c.size());
  for (size_t row = 0; row < c.size(); ++row) {
    double a_ = rho;
    double b_ = rho * c[row] - mu;
    double c_ = -mu * c[row] / 2.0;
    n[row] = (-b_ + std::sqrt(b_ * b_ - 4.0 * a_ * c_)) / (2.0 * a_);
    p[row] = c[row] + n[row];
  }
  return {p, n};
}

inline CSC resized(const CSC& a, int rows, int cols) {
  CSC m(rows, cols);
  for (int c = 0; c < cols; ++c) {
    if (c < a.cols)
      for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
        m.rowidx.push_back(a.rowidx[p]);
        m.val.push_back(a.val[p]);
      }
    m.colptr[c + 1] = static_cast<int>(m.rowidx.size());
  }
  return m;
}

// feasibility_restoration.hpp:347-628 (interior-point variant)
inline ExitStatus feasibility_restoration(const MatrixCallbacks& matrices,
                                          std::vector<IterationCallback>& iteration_callbacks,
                                          const Options& options, Vec& x, Vec& s, Vec& y, Vec& z,
                                          double mu, int& iterations, SolveStats* stats) {
  const int num_vars = matrices.num_decision_variables;
  const int num_eq = matrices.num_equality_constraints;
  const int num_ineq = matrices.num_inequality_constraints;
  constexpr double rho = 1e3;

  const Vec c_e = matrices.c_e(x);
  const Vec c_i = matrices.c_i(x);
  double fr_mu = std::max({mu, norm_inf(c_e), norm_inf(vsub(c_i, s))});
  const double zeta = std::sqrt(fr_mu);

  const Vec x_r = x;
  auto [p_e_0, n_e_0] = compute_p_n(c_e, rho, fr_mu);
  auto [p_i_0, n_i_0] = compute_p_n(vsub(c_i, s), rho, fr_mu);

  Vec D_r(num_vars);
  for (int i = 0; i < num_vars; ++i) D_r[i] = std::min(1.0 / (x[i] * x[i]), 1.0);

  const int nx = num_vars + 2 * num_eq + 2 * num_ineq;
  const int nz = num_ineq + 2 * num_eq + 2 * num_ineq;
  Vec fr_x;
  fr_x.reserve(nx);
  for (const Vec* v : {&x, &p_e_0, &n_e_0, &p_i_0, &n_i_0}) fr_x.insert(fr_x.end(), v->begin(), v->end());
  Vec fr_s(nz, 1.0);
  for (int i = 0; i < num_ineq; ++i) fr_s[i] = s[i];
  Vec fr_y(num_eq, 0.0);
  Vec fr_z;
  fr_z.reserve(nz);
  for (int i = 0; i < num_ineq; ++i) fr_z.push_back(fr_mu * (1.0 / s[i]));
  for (const Vec* v : {&p_e_0, &n_e_0, &p_i_0, &n_i_0})
    for (double e : *v) fr_z.push_back(fr_mu * (1.0 / e));

  Vec fr_d_c_i(nz, 1.0);
  for (int i = 0; i < num_ineq; ++i) fr_d_c_i[i] = matrices.scaling.c_i[i];
  ProblemScaling fr_scaling{1.0, matrices.scaling.c_e, fr_d_c_i};

  MatrixCallbacks fr;
  fr.num_decision_variables = nx;
  fr.num_equality_constraints = num_eq;
  fr.num_inequality_constraints = nz;
  fr.scaling = fr_scaling;
  auto head = [&](const Vec& xp) { return Vec(xp.begin(), xp.begin() + num_vars); };
  fr.f = [&, head](const Vec& xp) {
    double sum = 0.0;
    for (int i = num_vars; i < nx; ++i) sum += xp[i];
    double q = 0.0;
    for (int i = 0; i < num_vars; ++i) {
      double d = xp[i] - x_r[i];
      q += d * D_r[i] * d;
    }
    return rho * sum + zeta / 2.0 * q;
  };
  fr.g = [&](const Vec& xp) {
    Vec g(nx, rho);
    for (int i = 0; i < num_vars; ++i) g[i] = zeta * D_r[i] * (xp[i] - x_r[i]);
    return g;
  };
  fr.H = [&, head](const Vec& xp, const Vec& yp, const Vec& zp) {
    Vec d(nx, 0.0);
    std::vector<Triplet> t;
    for (int i = 0; i < num_vars; ++i) t.push_back({i, i, zeta * D_r[i]});
    CSC d2f = from_triplets(nx, nx, t);
    Vec zz(zp.begin(), zp.begin() + num_ineq);
    CSC H_c = resized(matrices.H_c(head(xp), yp, zz), nx, nx);
    return add(d2f, H_c);
  };
  fr.H_c = [&](const Vec&, const Vec&, const Vec&) { return CSC(nx, nx); };
  fr.c_e = [&, head](const Vec& xp) {
    Vec c = matrices.c_e(head(xp));
    for (int i = 0; i < num_eq; ++i) c[i] = c[i] - xp[num_vars + i] + xp[num_vars + num_eq + i];
    return c;
  };
  fr.A_e = [&, head](const Vec& xp) {
    CSC A_e = matrices.A_e(head(xp));
    std::vector<Triplet> t;
    append_as_triplets(t, 0, 0, {&A_e});
    append_diagonal_as_triplets(t, 0, num_vars, Vec(num_eq, -1.0));
    append_diagonal_as_triplets(t, 0, num_vars + num_eq, Vec(num_eq, 1.0));
    return from_triplets(A_e.rows, nx, t);
  };
  fr.c_i = [&, head](const Vec& xp) {
    Vec ci = matrices.c_i(head(xp));
    Vec out(nz);
    for (int i = 0; i < num_ineq; ++i)
      out[i] = ci[i] - xp[num_vars + 2 * num_eq + i] + xp[num_vars + 2 * num_eq + num_ineq + i];
    for (int i = 0; i < 2 * num_eq + 2 * num_ineq; ++i) out[num_ineq + i] = xp[num_vars + i];
    return out;
  };
  fr.A_i = [&, head](const Vec& xp) {
    CSC A_i = matrices.A_i(head(xp));
    std::vector<Triplet> t;
    append_as_triplets(t, 0, 0, {&A_i});
    append_diagonal_as_triplets(t, num_ineq, num_vars, Vec(2 * num_eq, 1.0));
    // columns of p_i: -I in the c_i rows, +I in the p_i >= 0 rows
    for (int i = 0; i < num_ineq; ++i) {
      t.push_back({i, num_vars + 2 * num_eq + i, -1.0});
      t.push_back({num_ineq + 2 * num_eq + i, num_vars + 2 * num_eq + i, 1.0});
    }
    for (int i = 0; i < num_ineq; ++i) {
      t.push_back({i, num_vars + 2 * num_eq + num_ineq + i, 1.0});
      t.push_back({num_ineq + 2 * num_eq + num_ineq + i, num_vars + 2 * num_eq + num_ineq + i, 1.0});
    }
    return from_triplets(2 * num_eq + 3 * num_ineq, nx, t);
  };

  auto status = interior_point(fr, iteration_callbacks, options, true, fr_x, fr_s, fr_y, fr_z,
                               fr_mu, iterations, stats);

  x.assign(fr_x.begin(), fr_x.begin() + num_vars);
  s.assign(fr_s.begin(), fr_s.begin() + num_ineq);

  if (status == ExitStatus::CALLBACK_REQUESTED_STOP) {
    Vec g = matrices.g(x);
    CSC A_e = matrices.A_e(x);
    CSC A_i = matrices.A_i(x);
    auto [ye, ze] = lagrange_multiplier_estimate(g, A_e, A_i, s, mu);
    y = ye;
    z = ze;
    return ExitStatus::SUCCESS;
  } else if (status == ExitStatus::SUCCESS) {
    return ExitStatus::LOCALLY_INFEASIBLE;
  } else {
    return ExitStatus::FEASIBILITY_RESTORATION_FAILED;
  }
}

}  // namespace orc
