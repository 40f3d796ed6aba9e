#include "ipm.hpp"

#include "ipm_decide.h"
#include "restoration.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace slpx {

namespace {

using Vec = std::vector<double>;
using clk = std::chrono::steady_clock;

double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

double norm_inf(const double* v, int n) {
  double m = 0.0;
  for (int i = 0; i < n; ++i) m = std::max(m, std::abs(v[i]));
  return m;
}
double norm_1(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += std::abs(v[i]);
  return s;
}
bool all_finite(const double* v, int n) {
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(v[i])) return false;
  return true;
}

// out += scale_rows(A)ᵀ v  with A in CSC over `vals`; row_scale may be null
void add_At_v(const CscPattern& A, const double* vals, const double* row_scale, const double* v,
              double sign, Vec& out) {
  for (int c = 0; c < A.cols; ++c) {
    double acc = 0.0;
    for (int p = A.colptr[c]; p < A.colptr[c + 1]; ++p) {
      const int r = A.rowidx[p];
      acc += (row_scale ? row_scale[r] * vals[p] : vals[p]) * v[r];
    }
    out[c] += sign * acc;
  }
}

enum class ErrType { INF_NORM_SCALED, ONE_NORM };

// Views into one downloaded V (see nlp.hpp for the layout)
struct VView {
  const NlpStructure& s;
  const Vec& V;
  double f() const { return V[s.off_f]; }
  const double* c_e() const { return V.data() + s.off_ce; }
  const double* c_i() const { return V.data() + s.off_ci; }
  const double* Ae() const { return V.data() + s.off_Ae; }
  const double* Ai() const { return V.data() + s.off_Ai; }
  Vec g_dense() const {
    Vec g(s.n, 0.0);
    for (int c = 0; c < s.n; ++c)
      for (int p = s.g_pat.colptr[c]; p < s.g_pat.colptr[c + 1]; ++p) g[c] += V[s.off_g + p];
    return g;
  }
};

// util/kkt_error.hpp:92-146.  `inv` = optional un-scaling (kkt_error.hpp:216-251):
// inv_f multiplies g, inv_ce/inv_ci multiply the rows of A_e/A_i and c_e/c_i/s,
// and y, z are replaced by d_c∘y·inv_f, d_c∘z·inv_f, μ by inv_f·μ.
template <ErrType T>
double kkt_error_impl(const NlpStructure& st, const Vec& g, const double* Ae, const double* c_e,
                      const double* Ai, const double* c_i, const Vec& s, const Vec& y, const Vec& z,
                      double mu, const Vec* scales) {
  const int n = st.n, m_e = st.m_e, m_i = st.m_i;
  const bool unscale = scales != nullptr;
  const double inv_f = unscale ? 1.0 / (*scales)[0] : 1.0;
  Vec inv_ce, inv_ci, yu(y), zu(z), su(s), ceu(m_e), ciu(m_i);
  if (unscale) {
    inv_ce.resize(m_e);
    inv_ci.resize(m_i);
    for (int j = 0; j < m_e; ++j) inv_ce[j] = 1.0 / (*scales)[1 + j];
    for (int j = 0; j < m_i; ++j) inv_ci[j] = 1.0 / (*scales)[1 + m_e + j];
    for (int j = 0; j < m_e; ++j) yu[j] = (*scales)[1 + j] * y[j] * inv_f;
    for (int j = 0; j < m_i; ++j) zu[j] = (*scales)[1 + m_e + j] * z[j] * inv_f;
    for (int j = 0; j < m_i; ++j) su[j] = inv_ci[j] * s[j];
  }
  for (int j = 0; j < m_e; ++j) ceu[j] = unscale ? inv_ce[j] * c_e[j] : c_e[j];
  for (int j = 0; j < m_i; ++j) ciu[j] = unscale ? inv_ci[j] * c_i[j] : c_i[j];
  const double muu = inv_f * mu;
  Vec dual(n);
  for (int i = 0; i < n; ++i) dual[i] = inv_f * g[i];
  add_At_v(st.Ae, Ae, unscale ? inv_ce.data() : nullptr, yu.data(), -1.0, dual);
  add_At_v(st.Ai, Ai, unscale ? inv_ci.data() : nullptr, zu.data(), -1.0, dual);
  Vec comp(m_i), cis(m_i);
  for (int j = 0; j < m_i; ++j) {
    comp[j] = su[j] * zu[j] - muu;
    cis[j] = ciu[j] - su[j];
  }
  if constexpr (T == ErrType::INF_NORM_SCALED) {
    constexpr double s_max = 100.0;
    const double s_d =
        std::max(s_max, (norm_1(yu.data(), m_e) + norm_1(zu.data(), m_i)) / double(m_e + m_i)) / s_max;
    const double s_c = std::max(s_max, norm_1(zu.data(), m_i) / double(m_i)) / s_max;
    return std::max({norm_inf(dual.data(), n) / s_d, norm_inf(comp.data(), m_i) / s_c,
                     norm_inf(ceu.data(), m_e), norm_inf(cis.data(), m_i)});
  } else {
    return norm_1(dual.data(), n) + norm_1(comp.data(), m_i) + norm_1(ceu.data(), m_e) +
           norm_1(cis.data(), m_i);
  }
}

bool scaling_is_identity(const NlpStructure& st, const Vec& scales) {
  // problem_scaling.hpp:111-113
  return scales[0] == 1.0 && st.m_e == 0 && st.m_i == 0;
}

// filter.hpp:17-212 (the entry type and the device's copy of the table: ipm_decide.h)
FilterEntry make_entry(double c, double v) { return FilterEntry{c, v}; }
FilterEntry make_entry(double f, const Vec& s, const double* c_e, int m_e, const double* c_i, double mu) {
  double logsum = 0.0, viol = norm_1(c_e, m_e);
  for (size_t j = 0; j < s.size(); ++j) {
    logsum += std::log(s[j]);
    viol += std::abs(c_i[j] - s[j]);
  }
  return FilterEntry{f - mu * logsum, viol};
}
bool dominated_by(const FilterEntry& a, const FilterEntry& e) { return filter_dominated_by(a, e); }

class Filter {
 public:
  double min_constraint_violation, max_constraint_violation;
  explicit Filter(double initial) {
    min_constraint_violation = 1e-4 * std::max(1.0, initial);
    max_constraint_violation = 1e4 * std::max(1.0, initial);
  }
  void reset() {
    m_filter.clear();
    m_last_rejection_due_to_filter = false;
  }
  bool try_add(const FilterEntry& cur, const FilterEntry& trial, double D_phi, double alpha) {
    // the rules: ipm_decide.h (one source for this driver and for the launch that decides the common iteration)
    FilterEntry add;
    bool insert = false;
    int last = m_last_rejection_due_to_filter ? 1 : 0;
    const int through = filter_rules(min_constraint_violation, max_constraint_violation, &last, cur, trial, D_phi, alpha,
                                     filter_powers(cur, D_phi, alpha), &add, &insert);
    m_last_rejection_due_to_filter = last != 0;
    if (!through) return false;
    for (auto& e : m_filter)
      if (dominated_by(trial, e)) {
        m_last_rejection_due_to_filter = true;
        return false;
      }
    if (insert) {
      m_filter.erase(std::remove_if(m_filter.begin(), m_filter.end(),
                                    [&](const FilterEntry& e) { return dominated_by(e, add); }),
                     m_filter.end());
      m_filter.push_back(add);
    }
    return true;
  }
  bool last_rejection_due_to_filter() const { return m_last_rejection_due_to_filter; }
  // the table as the device keeps it (ipm_decide.h); false: more entries than it holds
  bool to_state(FilterState& F) const {
    if (m_filter.size() > static_cast<size_t>(kFilterCapacity)) return false;
    F.min_constraint_violation = min_constraint_violation;
    F.max_constraint_violation = max_constraint_violation;
    F.n = static_cast<int>(m_filter.size());
    F.last_rejection_due_to_filter = m_last_rejection_due_to_filter ? 1 : 0;
    for (int k = 0; k < F.n; ++k) {
      F.ent[2 * k] = m_filter[k].cost;
      F.ent[2 * k + 1] = m_filter[k].constraint_violation;
    }
    return true;
  }
  void from_state(const FilterState& F) {
    min_constraint_violation = F.min_constraint_violation;
    max_constraint_violation = F.max_constraint_violation;
    m_last_rejection_due_to_filter = F.last_rejection_due_to_filter != 0;
    m_filter.clear();
    for (int k = 0; k < F.n; ++k) m_filter.push_back(FilterEntry{F.ent[2 * k], F.ent[2 * k + 1]});
  }

 private:
  std::vector<FilterEntry> m_filter;
  bool m_last_rejection_due_to_filter = false;
};

// fraction_to_the_boundary_rule.hpp:19-43
double ftb(const Vec& x, const Vec& p, double tau) {
  double alpha = 1.0;
  for (size_t i = 0; i < x.size(); ++i)
    if (alpha * p[i] < -tau * x[i]) alpha = -tau / p[i] * x[i];
  return alpha;
}

Vec axpy(const Vec& a, double alpha, const Vec& b) {
  Vec r(a.size());
  for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] + alpha * b[i];
  return r;
}

}  // namespace

std::vector<double> compute_problem_scaling(const NlpStructure& s, const std::vector<double>& V) {
  constexpr double g_max = 100.0;
  std::vector<double> scales(s.n_scales(), 1.0);
  double gn = 0.0;
  for (int p = 0; p < s.g_pat.nnz(); ++p) gn = std::max(gn, std::abs(V[s.off_g + p]));
  scales[0] = std::min(1.0, g_max / gn);
  std::vector<double> rn(s.m_e, 0.0);
  for (int p = 0; p < s.Ae.nnz(); ++p) rn[s.Ae.rowidx[p]] = std::max(rn[s.Ae.rowidx[p]], std::abs(V[s.off_Ae + p]));
  for (int j = 0; j < s.m_e; ++j) scales[1 + j] = std::min(g_max / rn[j], 1.0);
  rn.assign(s.m_i, 0.0);
  for (int p = 0; p < s.Ai.nnz(); ++p) rn[s.Ai.rowidx[p]] = std::max(rn[s.Ai.rowidx[p]], std::abs(V[s.off_Ai + p]));
  for (int j = 0; j < s.m_i; ++j) scales[1 + s.m_e + j] = std::min(g_max / rn[j], 1.0);
  return scales;
}

namespace {

ExitStatus feasibility_restoration(NewtonSystem& outer, const Vec& scales, const std::vector<IterationCallback>& user_callbacks,
                                   const std::function<bool(const FilterEntry&, double)>& outer_accepts, const Options& options,
                                   Vec& x, Vec& s, Vec& y, Vec& z, double mu, int& iterations, SolveReport& rep,
                                   clk::time_point solve_start, const Vec& c_e, const Vec& c_i, const Vec& g, double initial_violation);

// interior_point.hpp:129-878: the iteration proper, on caller-owned iterates (the
// restoration phase re-enters it with in_feasibility_restoration = true).
ExitStatus ipm_core_host(NewtonSystem& sys, const Vec& scales,
                         const std::vector<IterationCallback>& callbacks, const Options& options,
                         bool in_feasibility_restoration, Vec& x, Vec& s, Vec& y, Vec& z, double& mu,
                         int& iterations, SolveReport& rep, clk::time_point solve_start) {
  const NlpStructure& st = sys.structure();
  DeviceNlp& dev = sys.device();
  const int n = st.n, m_e = st.m_e, m_i = st.m_i, dim = n + m_e;
  auto finish = [&](ExitStatus st_) { return st_; };

  sys.reset_regularization();
  sys.set_gamma_min(in_feasibility_restoration ? 0.0 : 1e-10);  // :350-352

  Vec V(st.nV), Vtrial(st.nV);
  auto refresh_full = [&](const Vec& xx, const Vec& yy, const Vec& zz) {
    dev.upload_x(xx.data());
    dev.upload_duals(s.data(), yy.data(), zz.data());
    dev.sweep_full();
    dev.download_V(V.data());
  };
  // forward-only sweep at a trial point; fills the f, c_e, c_i head of Vtrial
  auto eval_values = [&](const Vec& xx) {
    dev.upload_x(xx.data());
    dev.sweep_values();
    dev.download(dev.d_V(), Vtrial.data(), static_cast<size_t>(st.off_g));
    ++rep.value_sweeps;
  };

  auto t_setup = clk::now();
  refresh_full(x, y, z);  // :245-251
  VView cur{st, V};
  Vec g = cur.g_dense();

  if (m_e > n) return finish(ExitStatus::TOO_FEW_DOFS);  // :274
  if (!all_finite(V.data(), st.nV)) return finish(ExitStatus::NONFINITE_INITIAL_GUESS);  // :283-286

  const double mu_min = scales[0] * options.tolerance / 10.0;  // :294
  constexpr double tau_min = 0.99;
  double tau = tau_min;

  double f = cur.f();
  Vec c_e(cur.c_e(), cur.c_e() + m_e), c_i(cur.c_i(), cur.c_i() + m_i);

  auto violation = [&](const Vec& ce, const Vec& ci, const Vec& ss) {
    double v = norm_1(ce.data(), m_e);
    for (int j = 0; j < m_i; ++j) v += std::abs(ci[j] - ss[j]);
    return v;
  };
  Filter filter{violation(c_e, c_i, s)};  // :303

  auto update_barrier = [&] {  // :308-333
    mu = std::max(mu_min, std::min(0.2 * mu, std::pow(mu, 1.5)));
    tau = std::max(tau_min, 1.0 - mu);
    filter.reset();
  };

  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-7;
  int full_step_rejected_counter = 0;
  const bool identity = scaling_is_identity(st, scales);
  auto E0_of = [&](const Vec& gg, const Vec& ce, const Vec& ci, const Vec& ss, const Vec& yy,
                   const Vec& zz) {
    return kkt_error_impl<ErrType::INF_NORM_SCALED>(st, gg, cur.Ae(), ce.data(), cur.Ai(),
                                                    ci.data(), ss, yy, zz, 0.0,
                                                    identity ? nullptr : &scales);
  };
  double E_0 = E0_of(g, c_e, c_i, s, y, z);  // :361-362
  rep.t_setup = since(t_setup);

  Vec p(dim), p_x(n), p_y(m_e), p_s(m_i), p_z(m_i);
  Vec trial_x, trial_s, trial_y, trial_z, trial_c_e(m_e), trial_c_i(m_i);
  double trial_f = 0.0;

  while (E_0 > options.tolerance) {
    // :387-408 infeasibility / divergence checks
    if (m_e > 0) {
      Vec t(n, 0.0);
      add_At_v(st.Ae, cur.Ae(), nullptr, c_e.data(), 1.0, t);
      double nt = 0.0, nc = 0.0;
      for (double v : t) nt += v * v;
      for (double v : c_e) nc += v * v;
      if (std::sqrt(nt) < 1e-6 && std::sqrt(nc) > 1e-2) return finish(ExitStatus::LOCALLY_INFEASIBLE);
    }
    if (m_i > 0) {
      Vec cplus(m_i), t(n, 0.0);
      for (int j = 0; j < m_i; ++j) cplus[j] = std::min(c_i[j], 0.0);
      add_At_v(st.Ai, cur.Ai(), nullptr, cplus.data(), 1.0, t);
      double nt = 0.0, nc = 0.0;
      for (double v : t) nt += v * v;
      for (double v : cplus) nc += v * v;
      if (std::sqrt(nt) < 1e-6 && std::sqrt(nc) > 1e-6) return finish(ExitStatus::LOCALLY_INFEASIBLE);
    }
    if (norm_inf(x.data(), n) > 1e10 || !all_finite(x.data(), n) || norm_inf(s.data(), m_i) > 1e10 ||
        !all_finite(s.data(), m_i))
      return finish(ExitStatus::DIVERGING_ITERATES);

    for (const auto& cb : callbacks)
      if (cb({iterations, x, s, y, z, V, &st, in_feasibility_restoration})) return finish(ExitStatus::CALLBACK_REQUESTED_STOP);

    // ---- Newton step on the device (:426-482) ----
    auto t0 = clk::now();
    dev.upload_x(x.data());
    dev.upload_duals(s.data(), y.data(), z.data());
    dev.upload_mu(&mu);
    dev.assemble();
    dev.build_rhs();
    rep.t_kkt_build += since(t0);  // enqueue time only: no host sync between the phases
    t0 = clk::now();
    // factorization attempts, each followed at once by solve + back-substitution
    auto info = sys.compute(/*solve_speculatively=*/true);
    rep.factorizations += sys.last_factorizations();
    rep.t_kkt_decomp += since(t0);
    if (info[0] != FactorInfo::Success) return finish(ExitStatus::FACTORIZATION_FAILED);  // :463-465
    rep.delta = sys.hessian_regularization()[0];
    rep.gamma = sys.constraint_jacobian_regularization()[0];

    auto download_step = [&](Vec& px, Vec& py, Vec& ps, Vec& pz) {
      dev.download(dev.d_p(), p.data(), dim);
      std::copy(p.begin(), p.begin() + n, px.begin());
      for (int j = 0; j < m_e; ++j) py[j] = -p[n + j];
      if (m_i) {
        dev.download(dev.d_ps(), ps.data(), m_i);
        dev.download(dev.d_pz(), pz.data(), m_i);
      }
    };
    t0 = clk::now();
    rep.solves += sys.last_factorizations();
    download_step(p_x, p_y, p_s, p_z);
    rep.t_kkt_solve += since(t0);

    t0 = clk::now();
    double alpha_max = ftb(s, p_s, tau);  // :488
    double alpha = alpha_max;
    bool call_feasibility_restoration = alpha < alpha_min;
    double alpha_z = ftb(z, p_z, tau);  // :497

    const FilterEntry current_entry = make_entry(f, s, c_e.data(), m_e, c_i.data(), mu);
    double D_phi = 0.0;  // :508-509
    for (int i = 0; i < n; ++i) D_phi += g[i] * p_x[i];
    {
      double t = 0.0;
      for (int j = 0; j < m_i; ++j) t += (1.0 / s[j]) * p_s[j];
      D_phi -= mu * t;
    }

    auto read_trial = [&] {
      trial_f = Vtrial[st.off_f];
      std::copy(Vtrial.begin() + st.off_ce, Vtrial.begin() + st.off_ce + m_e, trial_c_e.begin());
      std::copy(Vtrial.begin() + st.off_ci, Vtrial.begin() + st.off_ci + m_i, trial_c_i.begin());
    };

    while (true) {  // :512
      trial_x = axpy(x, alpha, p_x);
      eval_values(trial_x);
      read_trial();
      bool all_pos = true;
      for (double v : c_i) all_pos = all_pos && v > 0.0;
      if (options.feasible_ipm && all_pos) trial_s = trial_c_i;
      else trial_s = axpy(s, alpha, p_s);
      trial_y = axpy(y, alpha_z, p_y);
      trial_z = axpy(z, alpha_z, p_z);

      if (!std::isfinite(trial_f) || !all_finite(trial_c_e.data(), m_e) ||
          !all_finite(trial_c_i.data(), m_i)) {
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) {
          call_feasibility_restoration = true;
          break;
        }
        continue;
      }

      FilterEntry trial_entry = make_entry(trial_f, trial_s, trial_c_e.data(), m_e, trial_c_i.data(), mu);
      if (filter.try_add(current_entry, trial_entry, D_phi, alpha)) break;

      const double prev_violation = violation(c_e, c_i, s);
      double next_violation = violation(trial_c_e, trial_c_i, trial_s);

      // second-order corrections (:566-668): new rhs, SAME factorization
      if (alpha == alpha_max && next_violation >= prev_violation) {
        Vec soc_px = p_x, soc_ps = p_s, soc_py = p_y, soc_pz = p_z;
        double alpha_soc = alpha, alpha_z_soc = alpha_z;
        Vec c_e_soc = c_e, cims_soc(m_i);
        for (int j = 0; j < m_i; ++j) cims_soc[j] = c_i[j] - s[j];
        double soc_violation = next_violation;
        bool step_acceptable = false;
        for (int it = 0; it < 5 && !step_acceptable; ++it) {
          for (int j = 0; j < m_e; ++j) c_e_soc[j] = alpha_soc * c_e_soc[j] + trial_c_e[j];
          for (int j = 0; j < m_i; ++j) cims_soc[j] = alpha_soc * cims_soc[j] + trial_c_i[j] - trial_s[j];
          // rhs (:613-616) is O(nnz(A)) host work on the downloaded Jacobians
          Vec rhs(dim, 0.0);
          {
            Vec t(m_i);
            for (int j = 0; j < m_i; ++j) {
              const double sinv = 1.0 / s[j];
              t[j] = mu * sinv - (sinv * z[j]) * cims_soc[j];
            }
            for (int i = 0; i < n; ++i) rhs[i] = -g[i];
            add_At_v(st.Ae, cur.Ae(), nullptr, y.data(), 1.0, rhs);
            add_At_v(st.Ai, cur.Ai(), nullptr, t.data(), 1.0, rhs);
            for (int j = 0; j < m_e; ++j) rhs[n + j] = -c_e_soc[j];
          }
          SLPX_HIP_CHECK(hipMemcpyAsync(dev.d_rhs(), rhs.data(), dim * sizeof(double),
                                        hipMemcpyHostToDevice, dev.stream()));
          dev.solve();
          ++rep.solves;
          dev.download(dev.d_p(), p.data(), dim);
          std::copy(p.begin(), p.begin() + n, soc_px.begin());
          for (int j = 0; j < m_e; ++j) soc_py[j] = -p[n + j];
          {  // p_s, p_z with the corrected c_i - s (:479-480)
            Vec aipx(m_i, 0.0);
            for (int c = 0; c < n; ++c)
              for (int q = st.Ai.colptr[c]; q < st.Ai.colptr[c + 1]; ++q)
                aipx[st.Ai.rowidx[q]] += cur.Ai()[q] * soc_px[c];
            for (int j = 0; j < m_i; ++j) {
              const double sinv = 1.0 / s[j];
              soc_ps[j] = cims_soc[j] + aipx[j];
              soc_pz[j] = mu * sinv - z[j] - (sinv * z[j]) * soc_ps[j];
            }
          }
          alpha_soc = ftb(s, soc_ps, tau);
          alpha_z_soc = ftb(z, soc_pz, tau);
          trial_x = axpy(x, alpha_soc, soc_px);
          trial_s = axpy(s, alpha_soc, soc_ps);
          trial_y = axpy(y, alpha_z_soc, soc_py);
          trial_z = axpy(z, alpha_z_soc, soc_pz);
          eval_values(trial_x);
          read_trial();
          FilterEntry soc_entry = make_entry(trial_f, trial_s, trial_c_e.data(), m_e, trial_c_i.data(), mu);
          if (filter.try_add(current_entry, soc_entry, D_phi, alpha)) {
            p_x = soc_px;
            p_s = soc_ps;
            p_y = soc_py;
            p_z = soc_pz;
            alpha = alpha_soc;
            alpha_z = alpha_z_soc;
            step_acceptable = true;
            break;
          }
          next_violation = violation(trial_c_e, trial_c_i, trial_s);
          if (next_violation > 0.99 * soc_violation) break;
          soc_violation = next_violation;
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
      if (alpha < alpha_min) {  // :691-716
        const double current_kkt = kkt_error_impl<ErrType::ONE_NORM>(
            st, g, cur.Ae(), c_e.data(), cur.Ai(), c_i.data(), s, y, z, mu, nullptr);
        trial_x = axpy(x, alpha_max, p_x);
        trial_s = axpy(s, alpha_max, p_s);
        trial_y = axpy(y, alpha_z, p_y);
        trial_z = axpy(z, alpha_z, p_z);
        // needs g, A_e, A_i at the trial point: full sweep into a scratch copy
        Vec Vkeep = V;
        refresh_full(trial_x, trial_y, trial_z);
        Vec Vt = V;
        V = Vkeep;
        VView tv{st, Vt};
        trial_f = tv.f();
        std::copy(tv.c_e(), tv.c_e() + m_e, trial_c_e.begin());
        std::copy(tv.c_i(), tv.c_i() + m_i, trial_c_i.begin());
        const double next_kkt = kkt_error_impl<ErrType::ONE_NORM>(
            st, tv.g_dense(), tv.Ae(), trial_c_e.data(), tv.Ai(), trial_c_i.data(), trial_s,
            trial_y, trial_z, mu, nullptr);
        if (next_kkt <= 0.999 * current_kkt) break;
        call_feasibility_restoration = true;
        break;
      }
    }
    rep.t_line_search += since(t0);

    if (call_feasibility_restoration) {  // :721-771
      if (in_feasibility_restoration) return finish(ExitStatus::FEASIBILITY_RESTORATION_FAILED);

      const FilterEntry initial_entry = make_entry(f, s, c_e.data(), m_e, c_i.data(), mu);
      // Leave restoration once the outer filter accepts the restoration iterate and the violation dropped by 10 %
      // (:729-752); the restoration iteration reduces the outer problem's quantities at its iterate on the device
      auto outer_accepts = [&](const FilterEntry& trial_entry, double D_phi_restoration) {
        return filter.try_add(initial_entry, trial_entry, D_phi_restoration, alpha);
      };
      const ExitStatus fr_status = feasibility_restoration(sys, scales, callbacks, outer_accepts, options, x, s, y, z, mu, iterations,
                                                           rep, solve_start, c_e, c_i, g, initial_entry.constraint_violation);
      if (fr_status != ExitStatus::SUCCESS) return finish(fr_status);
      eval_values(x);
      read_trial();
      f = trial_f;
      c_e = trial_c_e;
      c_i = trial_c_i;
    } else {
      if (alpha == alpha_max) full_step_rejected_counter = 0;
      x = trial_x;
      s = trial_s;
      y = trial_y;
      z = trial_z;
      for (int j = 0; j < m_i; ++j) {  // :797-801
        constexpr double kappa = 1e10;
        z[j] = std::clamp(z[j], 1.0 / kappa * mu / s[j], kappa * mu / s[j]);
      }
      f = trial_f;
      c_e = trial_c_e;
      c_i = trial_c_i;
    }

    // AD refresh (:809-812)
    t0 = clk::now();
    refresh_full(x, y, z);
    g = cur.g_dense();
    rep.t_ad_refresh += since(t0);

    E_0 = E0_of(g, c_e, c_i, s, y, z);
    if (E_0 > options.tolerance) {  // :819-832
      auto E_mu_of = [&] {
        return kkt_error_impl<ErrType::INF_NORM_SCALED>(st, g, cur.Ae(), c_e.data(), cur.Ai(),
                                                        c_i.data(), s, y, z, mu, nullptr);
      };
      double E_mu = E_mu_of();
      while (mu > mu_min && E_mu <= 10.0 * mu) {
        update_barrier();
        E_mu = E_mu_of();
      }
    }
    if (options.diagnostics) {  // one line per iteration (print_iteration_diagnostics.hpp, condensed)
      std::fprintf(stderr,
                   "%4d  err %.3e  f %.6e  |c| %.3e  mu %.1e  delta %.3e  gamma %.3e  alpha %.2e  "
                   "alpha_z %.2e  nfact %d\n",
                   iterations, E_0, f, violation(c_e, c_i, s), mu, rep.delta, rep.gamma, alpha, alpha_z,
                   sys.last_factorizations());
    }
    ++iterations;
    rep.final_error = E_0;
    if (iterations >= options.max_iterations) return finish(ExitStatus::MAX_ITERATIONS_EXCEEDED);
    if (since(solve_start) > options.timeout) return finish(ExitStatus::TIMEOUT);
  }
  rep.final_error = E_0;
  return finish(ExitStatus::SUCCESS);
}

// The same iteration with the iterate, the step and every O(n) vector RESIDENT ON THE
// DEVICE (ipm_kernels.h): per iteration the host reads the
// inertia counters plus ~30 scalars (step sizes, directional derivative, filter entry of
// the trial point, error norms) and decides; it never sees a vector.  The kernels for the
// step sizes, the first trial point and its merit quantities are enqueued speculatively
// behind the Newton step, so the common iteration has two synchronizations: one after
// [step + first trial], one after [iterate update + AD refresh + error norms].
// Rare branches (the KKT-error fallback of the line search, feasibility restoration, user
// callbacks) pull the iterate to the host and reuse the host code above.
ExitStatus ipm_core_resident(NewtonSystem& sys, const Vec& scales,
                             const std::vector<IterationCallback>& callbacks, const Options& options,
                             bool in_feasibility_restoration, Vec& x, Vec& s, Vec& y, Vec& z, double& mu,
                             int& iterations, SolveReport& rep, clk::time_point solve_start) {
  const NlpStructure& st = sys.structure();
  DeviceNlp& dev = sys.device();
  const int n = st.n, m_e = st.m_e, m_i = st.m_i, dim = n + m_e;
  dev.ipm_enable();
  dev.ipm_set_error_scaling(scales);
  // (two sets of the scalar outputs, taken in turn by the iterations: the pipelined common iteration below enqueues
  // iteration k+1's launches before the host has read iteration k's)
  int slot = 0;
  dev.ipm_set_slot(slot);
  const IpmHost* Hp = &dev.ipm_host_slot(slot);
#define H (*Hp)

  sys.reset_regularization();
  sys.set_gamma_min(in_feasibility_restoration ? 0.0 : 1e-10);  // :350-352

  bool host_current = true;  // x, s, y, z on the host mirror the device iterate
  auto pull_state = [&] {
    if (host_current) return;
    dev.download(dev.d_x(), x.data(), n);
    if (m_i) {
      dev.download(dev.d_s(), s.data(), m_i);
      dev.download(dev.d_z(), z.data(), m_i);
    }
    if (m_e) dev.download(dev.d_y(), y.data(), m_e);
    host_current = true;
  };
  auto push_state = [&] {
    dev.upload_x(x.data());
    dev.upload_duals(s.data(), y.data(), z.data());
    host_current = true;
  };
  long twin_launches = 0, twin_taken = 0;
  bool spec_in_flight = false;  // the NEXT iteration's step and chain are enqueued already (and will run)
  bool spec_rides = false;      // ... and that step carries its own verdict (it runs whatever that is)
  long pipelined = 0, passed = 0;
  auto finish = [&](ExitStatus st_) {
    if (spec_in_flight) {  // (the step enqueued ahead runs for nothing: wait for it, forget it)
      dev.wait_published();
      sys.cancel_speculative_compute(/*launch_ran=*/spec_rides);
      spec_in_flight = false;
    }
    sys.set_after_attempt(nullptr);
    sys.set_twin_attempts(false);
    dev.ipm_set_slot(0);
    pull_state();
    if (std::getenv("SLPX_TWIN_VERBOSE")) {
      std::fprintf(stderr, "slpx pipelined iterations: %ld decided on the device, %ld steps enqueued ahead passed, of %d iterations\n",
                   pipelined, passed, iterations);
      const long* h = sys.twin_histogram();
      std::fprintf(stderr, "slpx twin attempts: %ld launches held two attempts, the policy took the second of %ld (%d factorizations, %d iterations); "
                   "first attempts of this system so far: %ld accepted, %ld / %ld with the failure the second stood for and the second accepted / not, "
                   "%ld with zero pivots, %ld with the other inertia failure, %ld failed\n",
                   twin_launches, twin_taken, rep.factorizations, iterations, h[0], h[1], h[2], h[3], h[4], h[5]);
    }
    return st_;
  };

  auto t_setup = clk::now();
  push_state();
  dev.upload_mu(&mu);
  double mu_on_device = mu;
  dev.sweep_full(/*with_reduce=*/false);  // :245-251 (the separable sums ride in the error launch)
  dev.ipm_errors(/*check_all_V=*/true, /*sums_ride=*/true);
  dev.wait_published();
  IpmErrOut cur = H.err;  // scalars of the current iterate

  if (m_e > n) return finish(ExitStatus::TOO_FEW_DOFS);                        // :274
  if (cur.finite == 0.0) return finish(ExitStatus::NONFINITE_INITIAL_GUESS);  // :283-286

  const double mu_min = scales[0] * options.tolerance / 10.0;  // :294
  constexpr double tau_min = 0.99;
  double tau = tau_min;
  Filter filter{cur.viol};  // :303
  auto update_barrier = [&] {  // :308-333
    mu = std::max(mu_min, std::min(0.2 * mu, std::pow(mu, 1.5)));
    tau = std::max(tau_min, 1.0 - mu);
    filter.reset();
  };

  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-7;
  constexpr double s_max = 100.0;
  int full_step_rejected_counter = 0;
  const bool identity = scaling_is_identity(st, scales);
  const char* lookahead_env = std::getenv("SLPX_IPM_LOOKAHEAD");
  const bool lookahead = lookahead_env == nullptr || lookahead_env[0] != '0';
  // util/kkt_error.hpp:92-146 from the reduced scalars
  auto E_mu_of = [&](const IpmErrOut& e, double m) {
    const double s_d = std::max(s_max, (e.y1 + e.z1) / double(m_e + m_i)) / s_max;
    const double s_c = std::max(s_max, e.z1 / double(m_i)) / s_max;
    const double comp = m_i ? std::max(std::abs(e.sz_max - m), std::abs(e.sz_min - m)) : 0.0;
    return std::max({e.dual_inf / s_d, comp / s_c, e.ce_inf, e.cis_inf});
  };
  auto E0_of = [&](const IpmErrOut& e) {
    if (identity) return E_mu_of(e, 0.0);
    const double s_d = std::max(s_max, (e.y1_u + e.z1_u) / double(m_e + m_i)) / s_max;
    const double s_c = std::max(s_max, e.z1_u / double(m_i)) / s_max;
    return std::max({e.dual_inf_u / s_d, e.sz_max_u / s_c, e.ce_inf_u, e.cis_inf_u});
  };
  double E_0 = E0_of(cur);  // :361-362
  rep.t_setup = since(t_setup);

  // ---- the pipelined common iteration (ipm_decide.h) ----
  // Most iterations end the same way: the filter takes the full step the look-ahead launch evaluated, the error is
  // above the tolerance, the barrier parameter stays.  Those decisions are then taken ON THE DEVICE, by the launch that
  // reduces the look-ahead iterate's norms — and the next iteration's step, with its chain, is enqueued as soon as this
  // iteration's factorization is accepted, BEFORE the verdict exists.  The deciding launch RIDES in that step's launch
  // (DeviceNlp::ipm_ride_errors_in_next_step): the step factors beside it and holds its results back until the verdict
  // is in; if the host has to look (a rejected step, a barrier update, convergence, anything rare) the step has run for
  // nothing and what was enqueued behind it passes.  (SLPX_IPM_RIDE=0, or where the riding workgroups do not fit beside
  // the tasks: the deciding launch in front of the step, which reads the word it leaves and passes if told to.)
  // The host then only follows: one poll per iteration, no launch on the critical path.
  // SLPX_IPM_PIPELINE=0: every iteration decided by the host, as before.
  const char* pipeline_env = std::getenv("SLPX_IPM_PIPELINE");
  const bool pipeline_on = lookahead && callbacks.empty() && !options.feasible_ipm && !(pipeline_env && pipeline_env[0] == '0');
  const bool ride_on = pipeline_on && dev.ipm_ride_possible();  // (SLPX_IPM_RIDE=0: the error launch in front of the gated step)
  bool ctl_current = false;      // the device's copy of (filter, current iterate's entry, mu) is the host's
  bool filter_on_device = false; // ... and newer than the host's: the device took decisions since
  auto sync_filter_from_device = [&] {
    if (!filter_on_device) return;
    filter.from_state(dev.ipm_pipeline_fetch().filter);
    filter_on_device = false;
  };
  auto upload_ctl = [&]() -> bool {
    static thread_local IpmCtl c;
    if (!filter.to_state(c.filter)) return false;
    c.mu = mu;
    c.mu_min = mu_min;
    c.tolerance = options.tolerance;
    c.cur_f = cur.f;
    c.cur_logsum = cur.logsum;
    c.cur_viol = cur.viol;
    c.m_e = m_e;
    c.m_i = m_i;
    c.identity_scaling = identity ? 1 : 0;
    dev.ipm_pipeline_upload(c);
    ctl_current = true;
    return true;
  };

  // host copies for the rare branches
  Vec V, p(dim), p_x(n), p_y(m_e), p_s(m_i), p_z(m_i);
  auto pull_V = [&] {
    V.resize(st.nV);
    dev.download_V(V.data());
  };

  while (E_0 > options.tolerance) {
    // :387-408 infeasibility / divergence checks
    if (m_e > 0 && std::sqrt(cur.aetce_sq) < 1e-6 && std::sqrt(cur.ce_sq) > 1e-2)
      return finish(ExitStatus::LOCALLY_INFEASIBLE);
    if (m_i > 0 && std::sqrt(cur.aitcp_sq) < 1e-6 && std::sqrt(cur.cp_sq) > 1e-6)
      return finish(ExitStatus::LOCALLY_INFEASIBLE);
    if (cur.x_inf > 1e10 || cur.s_inf > 1e10 || cur.finite == 0.0) return finish(ExitStatus::DIVERGING_ITERATES);

    if (!callbacks.empty()) {
      pull_state();
      pull_V();
      for (const auto& cb : callbacks)
        if (cb({iterations, x, s, y, z, V, &st, in_feasibility_restoration})) return finish(ExitStatus::CALLBACK_REQUESTED_STOP);
      push_state();  // a callback may have used this system's device buffers (restoration does)
    }

    // ---- Newton step (:426-482), then speculatively: step sizes, first trial point ----
    auto t0 = clk::now();
    const bool s_from_ci = options.feasible_ipm && cur.ci_all_pos != 0.0;
    const bool ahead = lookahead && !s_from_ci;
    if (mu != mu_on_device) {
      dev.upload_mu(&mu);
      mu_on_device = mu;
    }
    const bool resumed = spec_in_flight;  // this iteration's first launch (and chain) were enqueued by the one before
    spec_in_flight = false;
    // will THIS iteration's chain decide?  (its state on the device must be the host's, or the device's own)
    bool deciding = pipeline_on && ahead && (resumed || ctl_current || upload_ctl());
    if (!deciding) ctl_current = false;
    if (!resumed) dev.build_kkt_for_step(/*with_reduce=*/false);
    // Look-ahead (DeviceNlp::ipm_lookahead): instead of only f, c_e, c_i at the first trial point, the
    // WHOLE next iterate the full step would give — updated s, y, z, the full tape at it, the error norms
    // of :809-832 — is computed speculatively behind the step kernel, in a second set of buffers.  Most
    // iterations take the first trial point: the iteration is then complete when these numbers arrive
    // (one host round trip, four launches), and nothing is computed twice.  The feasible-IPM option
    // derives the trial s from the trial c_i (:520-526): it keeps the trial-values chain below.
    // (the error launch of a deciding chain comes LATER: riding in the next step's launch — it decides whether that
    // step counts while the step factors — or on its own where no step is enqueued ahead)
    const bool errors_later = deciding && ride_on;
    sys.set_after_attempt([&] {
      if (ahead) {
        dev.ipm_lookahead(tau);
        dev.sweep_full_lookahead();
        if (errors_later) {
        } else if (deciding) dev.ipm_errors_deciding(/*sums_ride=*/true);
        else dev.ipm_errors(false, /*sums_ride=*/true, /*ahead=*/true);
      } else {
        dev.ipm_direction(tau);
        dev.sweep_values_trial();
        dev.ipm_trial_metrics(-1.0, s_from_ci);
      }
    });
    // (twin attempts, NewtonSystem::compute_twin: the look-ahead launch takes the direction of whichever of a
    // launch's two attempts the regularization policy takes — the other after_attempt chain does not)
    sys.set_twin_attempts(ahead);
    auto info = sys.compute(/*solve_speculatively=*/true);
    twin_launches += sys.last_twin_launches();
    twin_taken += sys.last_twin_taken();
    // ---- the next iteration's step, ahead of this one's verdict ----
    bool spec = false, rode = false;
    if (deciding && info[0] == FactorInfo::Success && iterations + 2 < options.max_iterations) {
      dev.ipm_accept_lookahead();  // (the roles the buffers have if the device takes the step; put back below if not)
      dev.ipm_set_slot(slot ^ 1);
      // (the chain of the step enqueued ahead decides too: the device's state is its own by then)
      if (errors_later) rode = dev.ipm_ride_errors_in_next_step(slot);
      spec = sys.begin_speculative_compute(/*gated=*/!rode);
      if (!spec) {
        rode = false;
        dev.ipm_accept_lookahead();
        dev.ipm_set_slot(slot);
      }
    }
    if (errors_later && !rode) {
      if (spec) throw std::logic_error("slpx: a step enqueued ahead of its own verdict");
      dev.ipm_errors_deciding(/*sums_ride=*/true);  // (nothing was enqueued ahead: the error launch on its own)
    }
    const unsigned long long seq_this = dev.seq_expected();  // a chain that holds its error launch ends with this publication
    sys.set_twin_attempts(false);
    sys.set_after_attempt(nullptr);
    bool device_took_it = false;
    if (rode) {
      device_took_it = dev.ipm_ride_wait(slot);  // (the riding launch's scalars are in when its verdict is)
    } else {
      dev.wait_published_until(seq_this);  // compute() returns when the inertia counters are in; the trial chain may still run
      device_took_it = spec && H.go != 0.0;
    }
    if (spec) {
      if (device_took_it) {
        // the device took this iteration's decisions (ipm_decide.h): the filter accepted the full step, the
        // error is above the tolerance, mu stays; the next step is on its way.  The host follows.
        ++pipelined;
        rep.factorizations += sys.last_factorizations();
        rep.solves += sys.last_factorizations();
        rep.value_sweeps += sys.last_factorizations();
        rep.t_kkt_decomp += since(t0);
        rep.delta = sys.hessian_regularization()[0];
        rep.gamma = sys.constraint_jacobian_regularization()[0];
        full_step_rejected_counter = 0;
        host_current = false;
        filter_on_device = true;
        cur = H.err_ahead;
        E_0 = E0_of(cur);
        if (options.diagnostics) {
          std::fprintf(stderr,
                       "%4d  err %.3e  f %.6e  |c| %.3e  mu %.1e  delta %.3e  gamma %.3e  alpha %.2e  "
                       "alpha_z %.2e  nfact %d\n",
                       iterations, E_0, cur.f, cur.viol, mu, rep.delta, rep.gamma, H.dir.alpha_max, H.dir.alpha_z,
                       sys.last_factorizations());
        }
        ++iterations;
        rep.final_error = E_0;
        slot ^= 1;
        Hp = &dev.ipm_host_slot(slot);
        spec_in_flight = true;
        spec_rides = rode;
        if (since(solve_start) > options.timeout) return finish(ExitStatus::TIMEOUT);
        continue;
      }
      // the host has to look: what was enqueued ahead passes (a gated step at once; a step that carried its own verdict
      // runs and holds its results back, the chain behind it passes) — as if it had never been launched.  No waiting for
      // any of it: what the host enqueues next runs behind it in the stream's order, and it writes nothing the host or a
      // later launch reads but its sequence numbers.
      ++passed;
      sys.cancel_speculative_compute(/*launch_ran=*/rode);  // (a step that carried its own verdict ran, and held its results back)
      dev.ipm_accept_lookahead();
      dev.ipm_set_slot(slot);
    }
    ctl_current = false;  // (whatever happens below changes the filter, the iterate or mu on the host's side)
    sync_filter_from_device();
    rep.factorizations += sys.last_factorizations();
    rep.solves += sys.last_factorizations();
    rep.value_sweeps += sys.last_factorizations();
    rep.t_kkt_decomp += since(t0);
    if (info[0] != FactorInfo::Success) return finish(ExitStatus::FACTORIZATION_FAILED);  // :463-465
    rep.delta = sys.hessian_regularization()[0];
    rep.gamma = sys.constraint_jacobian_regularization()[0];

    t0 = clk::now();
    const double alpha_max = H.dir.alpha_max;  // :488
    double alpha = alpha_max;
    double alpha_z = H.dir.alpha_z;  // :497
    const double D_phi = H.dir.D_phi;  // :508-509
    bool call_feasibility_restoration = alpha < alpha_min;
    const FilterEntry current_entry{cur.f - mu * cur.logsum, cur.viol};
    double alpha_commit = alpha;
    bool commit_s_from_ci = s_from_ci;
    bool have_trial = true;  // the speculative chain already evaluated alpha_max
    bool trial_is_ahead = ahead;  // ... as the complete look-ahead iterate
    bool took_lookahead = false;

    while (true) {  // :512
      if (!have_trial) {
        dev.ipm_trial_point(alpha);
        dev.sweep_values_trial();
        dev.ipm_trial_metrics(alpha, s_from_ci);
        dev.wait_published();
        ++rep.value_sweeps;
      }
      have_trial = false;
      alpha_commit = alpha;
      const bool from_ahead = trial_is_ahead;
      trial_is_ahead = false;
      IpmTrialOut tr = from_ahead ? IpmTrialOut{H.err_ahead.f, H.err_ahead.viol, H.err_ahead.logsum, H.err_ahead.finite} : H.trial;

      if (tr.finite == 0.0) {
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) {
          call_feasibility_restoration = true;
          break;
        }
        continue;
      }

      const FilterEntry trial_entry{tr.f - mu * tr.logsum, tr.viol};
      if (filter.try_add(current_entry, trial_entry, D_phi, alpha)) {
        took_lookahead = from_ahead;
        break;
      }

      const double prev_violation = cur.viol;
      double next_violation = tr.viol;

      // second-order corrections (:566-668): new rhs, SAME factorization, all on the device
      if (alpha == alpha_max && next_violation >= prev_violation) {
        dev.ipm_save_direction();
        double alpha_soc = alpha, alpha_z_soc = alpha_z;
        double soc_violation = next_violation;
        bool step_acceptable = false;
        for (int it = 0; it < 5 && !step_acceptable; ++it) {
          dev.ipm_soc_accumulate(alpha_soc, it == 0, it == 0 && s_from_ci);
          dev.ipm_soc_rhs();
          dev.solve();
          ++rep.solves;
          dev.ipm_soc_backsub();
          dev.ipm_direction(tau);  // step sizes of the corrected direction + its trial point
          dev.sweep_values_trial();
          dev.ipm_trial_metrics(-1.0, false);
          dev.wait_published();
          ++rep.value_sweeps;
          alpha_soc = H.dir.alpha_max;
          alpha_z_soc = H.dir.alpha_z;
          tr = H.trial;
          const FilterEntry soc_entry{tr.f - mu * tr.logsum, tr.viol};
          if (filter.try_add(current_entry, soc_entry, D_phi, alpha)) {
            alpha = alpha_soc;
            alpha_z = alpha_z_soc;
            alpha_commit = alpha_soc;
            commit_s_from_ci = false;
            step_acceptable = true;
            break;
          }
          next_violation = tr.viol;
          if (next_violation > 0.99 * soc_violation) break;
          soc_violation = next_violation;
        }
        if (step_acceptable) break;
        dev.ipm_restore_direction();
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
      if (alpha < alpha_min) {  // :691-716 — rare: on the host, with the iterate pulled over
        pull_state();
        pull_V();
        VView cv{st, V};
        const Vec g = cv.g_dense();
        const double current_kkt = kkt_error_impl<ErrType::ONE_NORM>(st, g, cv.Ae(), cv.c_e(), cv.Ai(),
                                                                     cv.c_i(), s, y, z, mu, nullptr);
        dev.download(dev.d_p(), p.data(), dim);
        std::copy(p.begin(), p.begin() + n, p_x.begin());
        for (int j = 0; j < m_e; ++j) p_y[j] = -p[n + j];
        if (m_i) {
          dev.download(dev.d_ps(), p_s.data(), m_i);
          dev.download(dev.d_pz(), p_z.data(), m_i);
        }
        const Vec trial_x = axpy(x, alpha_max, p_x), trial_s = axpy(s, alpha_max, p_s),
                  trial_y = axpy(y, alpha_z, p_y), trial_z = axpy(z, alpha_z, p_z);
        // needs g, A_e, A_i at the trial point: full sweep there, then put the iterate back
        dev.upload_x(trial_x.data());
        dev.upload_duals(s.data(), trial_y.data(), trial_z.data());
        dev.sweep_full();
        Vec Vt(st.nV);
        dev.download_V(Vt.data());
        push_state();
        // the device V must describe x again (feasibility_restoration.hpp:393-394 evaluates at
        // the current iterate): the restoration entry below, and the next step if the fallback
        // accepts, read it
        dev.sweep_full();
        VView tv{st, Vt};
        const double next_kkt = kkt_error_impl<ErrType::ONE_NORM>(st, tv.g_dense(), tv.Ae(), tv.c_e(), tv.Ai(),
                                                                  tv.c_i(), trial_s, trial_y, trial_z, mu,
                                                                  nullptr);
        if (next_kkt <= 0.999 * current_kkt) {
          alpha_commit = alpha_max;
          commit_s_from_ci = false;
          break;
        }
        call_feasibility_restoration = true;
        break;
      }
    }
    rep.t_line_search += since(t0);

    t0 = clk::now();
    if (call_feasibility_restoration) {  // :721-771 — rare: host vectors, as in ipm_core_host
      if (in_feasibility_restoration) return finish(ExitStatus::FEASIBILITY_RESTORATION_FAILED);
      pull_state();
      pull_V();
      VView cv{st, V};
      const Vec g = cv.g_dense();
      const Vec c_e(cv.c_e(), cv.c_e() + m_e), c_i(cv.c_i(), cv.c_i() + m_i);
      const double f = cv.f();
      const FilterEntry initial_entry = make_entry(f, s, c_e.data(), m_e, c_i.data(), mu);
      // Leave restoration once the outer filter accepts the restoration iterate and the
      // violation dropped by 10 % (:729-752).
      auto outer_accepts = [&](const FilterEntry& trial_entry, double D_phi_restoration) {
        return filter.try_add(initial_entry, trial_entry, D_phi_restoration, alpha);
      };
      const ExitStatus fr_status = feasibility_restoration(sys, scales, callbacks, outer_accepts, options, x, s, y, z, mu, iterations,
                                                           rep, solve_start, c_e, c_i, g, initial_entry.constraint_violation);
      if (fr_status != ExitStatus::SUCCESS) return finish(fr_status);
      push_state();
    } else {
      if (alpha == alpha_max) full_step_rejected_counter = 0;
      if (took_lookahead) {
        dev.ipm_accept_lookahead();  // :775-801 happened in ipm_lookahead_kernel; the buffers change roles
      } else {
        dev.ipm_commit(alpha_commit, alpha_z, commit_s_from_ci);  // :775-801
      }
      host_current = false;
    }

    // AD refresh (:809-812) and every norm the next decisions need
    if (took_lookahead && !call_feasibility_restoration) {
      cur = H.err_ahead;  // already there: the look-ahead chain swept and reduced this iterate
    } else {
      dev.sweep_full(/*with_reduce=*/false);
      dev.ipm_errors(false, /*sums_ride=*/true);
      dev.wait_published();
      cur = H.err;
    }
    rep.t_ad_refresh += since(t0);

    E_0 = E0_of(cur);
    if (E_0 > options.tolerance) {  // :819-832
      double E_mu = E_mu_of(cur, mu);
      while (mu > mu_min && E_mu <= 10.0 * mu) {
        update_barrier();
        E_mu = E_mu_of(cur, mu);
      }
    }
    if (options.diagnostics) {
      std::fprintf(stderr,
                   "%4d  err %.3e  f %.6e  |c| %.3e  mu %.1e  delta %.3e  gamma %.3e  alpha %.2e  "
                   "alpha_z %.2e  nfact %d\n",
                   iterations, E_0, cur.f, cur.viol, mu, rep.delta, rep.gamma, alpha, alpha_z,
                   sys.last_factorizations());
    }
    ++iterations;
    rep.final_error = E_0;
    if (iterations >= options.max_iterations) return finish(ExitStatus::MAX_ITERATIONS_EXCEEDED);
    if (since(solve_start) > options.timeout) return finish(ExitStatus::TIMEOUT);
  }
  rep.final_error = E_0;
  return finish(ExitStatus::SUCCESS);
}

#undef H

// SLPX_IPM_RESIDENT=0 keeps the O(n) logic between Newton steps on the host (the round-1
// arrangement; kept as the cross-check of the device-resident iteration).
ExitStatus ipm_core(NewtonSystem& sys, const Vec& scales, const std::vector<IterationCallback>& callbacks,
                    const Options& options, bool in_feasibility_restoration, Vec& x, Vec& s, Vec& y, Vec& z,
                    double& mu, int& iterations, SolveReport& rep, clk::time_point solve_start) {
  const char* env = std::getenv("SLPX_IPM_RESIDENT");
  const bool resident = !(env && env[0] == '0');
  return (resident ? ipm_core_resident : ipm_core_host)(sys, scales, callbacks, options,
                                                        in_feasibility_restoration, x, s, y, z, mu, iterations,
                                                        rep, solve_start);
}

// feasibility_restoration.hpp:26-101: p, n >= 0 with p - n = c minimising the barrier
// problem  rho (p + n) - mu (ln p + ln n)
void compute_p_n(const Vec& c, double rho, double mu, Vec& p, Vec& n) {
  p.resize(c.size());
  n.resize(c.size());
  for (size_t row = 0; row < c.size(); ++row) {
    const double a_ = rho, b_ = rho * c[row] - mu, c_ = -mu * c[row] / 2.0;
    n[row] = (-b_ + std::sqrt(b_ * b_ - 4.0 * a_ * c_)) / (2.0 * a_);
    p[row] = c[row] + n[row];
  }
}

// (the restoration problem itself is never built as a model: restoration.hpp — its extra variables are eliminated from
// the Newton-KKT system in closed form and the rest runs on the outer problem's own compiled system)

// util/lagrange_multiplier_estimate.hpp:56-133: least-squares (y, z) of
//   [A_e 0; A_i -S] [A_e 0; A_i -S]^T [y; z] = [A_e 0; A_i -S] [g; -mu 1].
// Eliminating the auxiliary unknowns of the equivalent equality-constrained QP gives a
// system with the KKT pattern this NewtonSystem already has a symbolic factorization for:
//   [I + A_i^T S^-2 A_i   A_e^T] [ d ]   [-g + mu A_i^T S^-1 1]
//   [A_e                  0    ] [-y ] = [0                   ],  z = mu S^-1 1 - S^-2 A_i d
// V must hold g, A_e, A_i at the current x.
bool lagrange_multiplier_estimate(NewtonSystem& sys, const Vec& V, const Vec& s, double mu, Vec& y,
                                  Vec& z, SolveReport& rep) {
  const NlpStructure& st = sys.structure();
  DeviceNlp& dev = sys.device();
  const int n = st.n, m_e = st.m_e, m_i = st.m_i, dim = n + m_e;
  VView cur{st, V};
  Vec rhs(dim, 0.0), sinv_mu(m_i);
  const Vec g = cur.g_dense();
  for (int i = 0; i < n; ++i) rhs[i] = -g[i];
  for (int j = 0; j < m_i; ++j) sinv_mu[j] = mu / s[j];
  add_At_v(st.Ai, cur.Ai(), nullptr, sinv_mu.data(), 1.0, rhs);

  Vec zeros_e(std::max(1, m_e), 0.0), ones_i(std::max(1, m_i), 1.0);
  dev.upload_duals(s.data(), zeros_e.data(), ones_i.data());
  dev.assemble_lsq();
  SLPX_HIP_CHECK(hipMemcpyAsync(dev.d_rhs(), rhs.data(), dim * sizeof(double), hipMemcpyHostToDevice,
                                dev.stream()));
  // The reference factors this system without any regularization (a SimplicialLDLT of its own,
  // :107): the top-left block I + A_i^T S^-2 A_i is positive definite, so delta = gamma = 0 is
  // tried first whatever the KKT pattern's structural zeros say — starting the usual loop at
  // delta = 1e-4 instead perturbed the estimate by 1e-4 relative (tests/test_restoration_gpu.py).
  // Only a rank-deficient A_e falls back to the regularizing loop, with its own history.
  // Only when that breaks down (a constraint row the elimination order could not pair with a
  // variable is a zero pivot without gamma; a rank-deficient A_e) the regularizing loop runs,
  // with its own history — and three steps of iterative refinement against the unregularized
  // matrix take its delta, gamma out of the answer again.
  bool regularized = false;
  if (sys.factor_unregularized()) {
    ++rep.factorizations;
  } else {
    const auto saved = sys.regularization_state();
    sys.reset_regularization();
    const auto info = sys.compute();
    rep.factorizations += 1 + sys.last_factorizations();
    sys.set_regularization_state(saved);
    if (info[0] != FactorInfo::Success) return false;
    regularized = true;
  }
  dev.solve();
  ++rep.solves;
  if (regularized) {
    dev.refine_solution(3);
    rep.solves += 3;
  }
  Vec p(dim);
  dev.download(dev.d_p(), p.data(), dim);
  y.assign(m_e, 0.0);
  for (int j = 0; j < m_e; ++j) y[j] = -p[n + j];
  Vec aid(m_i, 0.0);
  for (int c = 0; c < n; ++c)
    for (int q = st.Ai.colptr[c]; q < st.Ai.colptr[c + 1]; ++q) aid[st.Ai.rowidx[q]] += cur.Ai()[q] * p[c];
  z.assign(m_i, 0.0);
  for (int j = 0; j < m_i; ++j) {
    constexpr double kappa = 1e10;
    const double zj = mu / s[j] - aid[j] / (s[j] * s[j]);
    z[j] = std::clamp(zj, 1.0 / kappa * mu / s[j], kappa * mu / s[j]);  // :125-130
  }
  return true;
}

// The restoration problem as the user's iteration callbacks see it (slpx.h: slpx_iteration_info inside restoration):
// n + 2 m_e + 2 m_i variables [x | p_e | n_e | p_i | n_i], m_i + 2 m_e + 2 m_i inequality rows, the matrices of
// feasibility_restoration.hpp:434-593 as value arrays over static patterns.  Only built when a solve with user
// callbacks enters restoration; the solver itself never forms these.
struct RestorationView {
  NlpStructure st;
  std::vector<double> V;
};
void build_restoration_view(const NlpStructure& o, RestorationView& v) {
  NlpStructure& t = v.st;
  const int n = o.n, m_e = o.m_e, m_i = o.m_i, M = 2 * m_e + 2 * m_i;
  t.n = n + M;
  t.m_e = m_e;
  t.m_i = m_i + M;
  auto cols = [&](CscPattern& p, int rows) {
    p.rows = rows;
    p.cols = t.n;
    p.colptr.assign(1, 0);
    p.rowidx.clear();
  };
  auto close = [](CscPattern& p) { p.colptr.push_back(static_cast<int32_t>(p.rowidx.size())); };
  cols(t.g_pat, 1);
  for (int c = 0; c < t.n; ++c) {
    t.g_pat.rowidx.push_back(0);
    close(t.g_pat);
  }
  cols(t.Ae, m_e);
  cols(t.Ai, t.m_i);
  cols(t.Hf, t.n);
  cols(t.Hc, t.n);
  for (int c = 0; c < n; ++c) {
    for (int q = o.Ae.colptr[c]; q < o.Ae.colptr[c + 1]; ++q) t.Ae.rowidx.push_back(o.Ae.rowidx[q]);
    for (int q = o.Ai.colptr[c]; q < o.Ai.colptr[c + 1]; ++q) t.Ai.rowidx.push_back(o.Ai.rowidx[q]);
    t.Hf.rowidx.push_back(c);
    for (int q = o.Hc.colptr[c]; q < o.Hc.colptr[c + 1]; ++q) t.Hc.rowidx.push_back(o.Hc.rowidx[q]);
    close(t.Ae), close(t.Ai), close(t.Hf), close(t.Hc);
  }
  for (int e = 0; e < M; ++e) {  // columns of p_e, n_e, p_i, n_i (:499-573)
    if (e < 2 * m_e) t.Ae.rowidx.push_back(e % m_e);
    if (e >= 2 * m_e) t.Ai.rowidx.push_back((e - 2 * m_e) % m_i);
    t.Ai.rowidx.push_back(m_i + e);
    close(t.Ae), close(t.Ai), close(t.Hf), close(t.Hc);
  }
  t.off_f = 0;
  t.off_ce = 1;
  t.off_ci = t.off_ce + t.m_e;
  t.off_g = t.off_ci + t.m_i;
  t.off_Ae = t.off_g + t.g_pat.nnz();
  t.off_Ai = t.off_Ae + t.Ae.nnz();
  t.off_Hf = t.off_Ai + t.Ai.nnz();
  t.off_Hc = t.off_Hf + t.Hf.nnz();
  t.nV = t.off_Hc + t.Hc.nnz();
}
void fill_restoration_view(const NlpStructure& o, RestorationView& v, const Vec& Vo, const Vec& x_ext, const Vec& x_r, const Vec& w,
                           double rho) {
  const NlpStructure& t = v.st;
  const int n = o.n, m_e = o.m_e, m_i = o.m_i, M = 2 * m_e + 2 * m_i;
  Vec& V = v.V;
  V.assign(t.nV, 0.0);
  const double* pn = x_ext.data() + n;
  double f = 0.0;
  for (int e = 0; e < M; ++e) f += rho * pn[e];
  for (int k = 0; k < n; ++k) f += 0.5 * w[k] * (x_ext[k] - x_r[k]) * (x_ext[k] - x_r[k]);
  V[t.off_f] = f;
  for (int j = 0; j < m_e; ++j) V[t.off_ce + j] = Vo[o.off_ce + j] - pn[j] + pn[m_e + j];
  for (int r = 0; r < m_i; ++r) V[t.off_ci + r] = Vo[o.off_ci + r] - pn[2 * m_e + r] + pn[2 * m_e + m_i + r];
  for (int e = 0; e < M; ++e) V[t.off_ci + m_i + e] = pn[e];
  for (int k = 0; k < n; ++k) V[t.off_g + k] = w[k] * (x_ext[k] - x_r[k]);
  for (int e = 0; e < M; ++e) V[t.off_g + n + e] = rho;
  int qe = t.off_Ae, qi = t.off_Ai, qh = t.off_Hc;
  for (int c = 0; c < n; ++c) {
    for (int q = o.Ae.colptr[c]; q < o.Ae.colptr[c + 1]; ++q) V[qe++] = Vo[o.off_Ae + q];
    for (int q = o.Ai.colptr[c]; q < o.Ai.colptr[c + 1]; ++q) V[qi++] = Vo[o.off_Ai + q];
    V[t.off_Hf + c] = w[c];
    for (int q = o.Hc.colptr[c]; q < o.Hc.colptr[c + 1]; ++q) V[qh++] = Vo[o.off_Hc + q];
  }
  for (int e = 0; e < M; ++e) {
    const bool minus = e < m_e || (e >= 2 * m_e && e < 2 * m_e + m_i);  // p_e, p_i enter their constraint with -1
    if (e < 2 * m_e) V[qe++] = minus ? -1.0 : 1.0;
    else V[qi++] = minus ? -1.0 : 1.0;
    V[qi++] = 1.0;
  }
}

// kkt_error.hpp:92-146 (1-norm variant) of the restoration problem, on the host: the rare fallback of the line search
// (interior_point.hpp:691-716).  Vo = the OUTER problem's V at x; X = [x | p_e | n_e | p_i | n_i], S, Z = the five
// inequality blocks.
double restoration_kkt_error_one_norm(const NlpStructure& o, const Vec& Vo, const Vec& X, const Vec& S, const Vec& y, const Vec& Z,
                                      const Vec& x_r, const Vec& w, double rho, double mu) {
  const int n = o.n, m_e = o.m_e, m_i = o.m_i, M = 2 * m_e + 2 * m_i;
  const double* pn = X.data() + n;
  Vec dual(n + M, 0.0);
  for (int k = 0; k < n; ++k) dual[k] = w[k] * (X[k] - x_r[k]);
  for (int c = 0; c < n; ++c) {
    for (int q = o.Ae.colptr[c]; q < o.Ae.colptr[c + 1]; ++q) dual[c] -= Vo[o.off_Ae + q] * y[o.Ae.rowidx[q]];
    for (int q = o.Ai.colptr[c]; q < o.Ai.colptr[c + 1]; ++q) dual[c] -= Vo[o.off_Ai + q] * Z[o.Ai.rowidx[q]];
  }
  for (int j = 0; j < m_e; ++j) {
    dual[n + j] = rho + y[j] - Z[m_i + j];
    dual[n + m_e + j] = rho - y[j] - Z[m_i + m_e + j];
  }
  for (int r = 0; r < m_i; ++r) {
    dual[n + 2 * m_e + r] = rho + Z[r] - Z[m_i + 2 * m_e + r];
    dual[n + 2 * m_e + m_i + r] = rho - Z[r] - Z[m_i + 2 * m_e + m_i + r];
  }
  double err = norm_1(dual.data(), n + M);
  for (int j = 0; j < m_e; ++j) err += std::abs(Vo[o.off_ce + j] - pn[j] + pn[m_e + j]);
  for (int r = 0; r < m_i; ++r) {
    const double c = Vo[o.off_ci + r] - pn[2 * m_e + r] + pn[2 * m_e + m_i + r];
    err += std::abs(S[r] * Z[r] - mu) + std::abs(c - S[r]);
  }
  for (int e = 0; e < M; ++e) err += std::abs(S[m_i + e] * Z[m_i + e] - mu) + std::abs(pn[e] - S[m_i + e]);
  return err;
}

// interior_point.hpp:129-878 on the restoration problem (in_feasibility_restoration = true), the counterpart of
// ipm_core_resident for it: the iterate stays on the device — x, s_0, y, z_0 in the outer system's buffers, p, n and
// their slacks and duals in the FrDevice —, every Newton step is a factorization of the reduced system on the outer
// system's plan (NewtonSystem::compute_hooked), and the host decides on a few dozen scalars.  `accept_exit` is the
// callback the reference appends (interior_point.hpp:729-752): the outer filter's verdict on the restoration iterate,
// from the outer problem's quantities the error launch reduces along.
ExitStatus restoration_core(NewtonSystem& sys, FrDevice& fr, const std::vector<IterationCallback>& user_callbacks,
                            const std::function<bool(const FrErrOut&)>& accept_exit, const Options& options, const Vec& x_r, const Vec& w,
                            Vec& X, Vec& S, Vec& y, Vec& Z, double& mu, int& iterations, SolveReport& rep, clk::time_point solve_start) {
  const NlpStructure& st = sys.structure();
  DeviceNlp& dev = sys.device();
  const int n = st.n, m_e = st.m_e, m_i = st.m_i, M = 2 * m_e + 2 * m_i, dim = n + m_e, mi_ext = m_i + M;
  constexpr double rho = 1e3;
  const FrHost& H = fr.host();

  sys.reset_regularization();
  sys.set_gamma_min(0.0);  // :350-352

  bool host_current = true;
  auto pull_state = [&] {
    if (host_current) return;
    dev.download(dev.d_x(), X.data(), n);
    if (m_i) {
      dev.download(dev.d_s(), S.data(), m_i);
      dev.download(dev.d_z(), Z.data(), m_i);
    }
    if (m_e) dev.download(dev.d_y(), y.data(), m_e);
    fr.download_state(X.data() + n, S.data() + m_i, Z.data() + m_i);
    host_current = true;
  };
  auto finish = [&](ExitStatus st_) {
    pull_state();
    dev.state_changed_by_caller();
    return st_;
  };

  auto t_setup = clk::now();
  dev.sweep_full();  // :245-251: the outer tape at (x, y, z_0) is the restoration problem's c_e, c_i, A_e, A_i, H_c
  fr.errors(/*check_all_V=*/true, mu);
  fr.wait_published();
  FrErrOut cur = H.err;
  if (cur.e.finite == 0.0) return finish(ExitStatus::NONFINITE_INITIAL_GUESS);  // :283-286

  const double mu_min = options.tolerance / 10.0;  // :294 (the restoration cost is not scaled)
  constexpr double tau_min = 0.99;
  double tau = tau_min;
  Filter filter{cur.e.viol};  // :303
  auto update_barrier = [&] {  // :308-333
    mu = std::max(mu_min, std::min(0.2 * mu, std::pow(mu, 1.5)));
    tau = std::max(tau_min, 1.0 - mu);
    filter.reset();
  };
  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-7;
  constexpr double s_max = 100.0;
  int full_step_rejected_counter = 0;
  // util/kkt_error.hpp:92-146 from the reduced scalars (the scaling {1, d_ce, [d_ci, 1...]} is never the identity here)
  auto E_mu_of = [&](const IpmErrOut& e, double m) {
    const double s_d = std::max(s_max, (e.y1 + e.z1) / double(m_e + mi_ext)) / s_max;
    const double s_c = std::max(s_max, e.z1 / double(mi_ext)) / s_max;
    const double comp = std::max(std::abs(e.sz_max - m), std::abs(e.sz_min - m));
    return std::max({e.dual_inf / s_d, comp / s_c, e.ce_inf, e.cis_inf});
  };
  auto E0_of = [&](const IpmErrOut& e) {
    const double s_d = std::max(s_max, (e.y1_u + e.z1_u) / double(m_e + mi_ext)) / s_max;
    const double s_c = std::max(s_max, e.z1_u / double(mi_ext)) / s_max;
    return std::max({e.dual_inf_u / s_d, e.sz_max_u / s_c, e.ce_inf_u, e.cis_inf_u});
  };
  double E_0 = E0_of(cur.e);  // :361-362
  rep.t_setup += since(t_setup);

  RestorationView view;
  Vec Vo;
  bool previous_took_first_attempt = true;  // ... of the regularization policy

  while (E_0 > options.tolerance) {
    // :387-408
    if (m_e > 0 && std::sqrt(cur.e.aetce_sq) < 1e-6 && std::sqrt(cur.e.ce_sq) > 1e-2) return finish(ExitStatus::LOCALLY_INFEASIBLE);
    if (std::sqrt(cur.e.aitcp_sq) < 1e-6 && std::sqrt(cur.e.cp_sq) > 1e-6) return finish(ExitStatus::LOCALLY_INFEASIBLE);
    if (cur.e.x_inf > 1e10 || cur.e.s_inf > 1e10 || cur.e.finite == 0.0) return finish(ExitStatus::DIVERGING_ITERATES);

    if (!user_callbacks.empty()) {
      pull_state();
      Vo.resize(st.nV);
      dev.download_V(Vo.data());
      if (view.st.n == 0) build_restoration_view(st, view);
      fill_restoration_view(st, view, Vo, X, x_r, w, rho);
      for (const auto& cb : user_callbacks)
        if (cb({iterations, X, S, y, Z, view.V, &view.st, true})) return finish(ExitStatus::CALLBACK_REQUESTED_STOP);
    }
    if (accept_exit(cur)) return finish(ExitStatus::CALLBACK_REQUESTED_STOP);

    // ---- Newton step (:426-482) on the reduced system, then speculatively: the full direction, its step sizes, the
    // first trial point and its filter entry ----
    auto t0 = clk::now();
    // Two attempts of the regularization policy per launch (NewtonSystem::compute_hooked): the one it is at and the one
    // it would make next — a restoration phase rejects its unregularized attempt iteration after iteration (H_c with the
    // restoration's own multipliers is indefinite), and climbs the ladder from delta = 1e-4 (:95-98, :127-130) every few
    // iterations: each rejected attempt used to be a launch and a round trip of its own.  The chain behind a launch —
    // the look-ahead iteration, as in ipm_core_resident: the whole iterate the full step would give, the full tape at it,
    // its error norms — follows the FIRST attempt, and only where the previous iteration took its first attempt; a
    // direction taken from a second attempt gets its chain when the policy has settled.
    NewtonSystem::AttemptHooks hooks;
    auto chain = [&](double d) {
      fr.expand(d, mu, tau, /*soc=*/false, /*ahead=*/true);
      dev.sweep_full_lookahead(/*with_reduce=*/false, /*skippable=*/false);  // (the separable sums ride in the error launch)
      fr.errors(false, mu, /*ahead=*/true, /*sums_ride=*/true);
    };
    hooks.prepare = [&](double d, double) { fr.build(d, mu, /*soc=*/false, /*rhs_only=*/false); };
    hooks.prepare_second = [&](double d, double, const double** lhs2, const double** rhs2) {
      fr.build(d, mu, /*soc=*/false, /*rhs_only=*/false, /*second=*/true);
      *lhs2 = fr.second_lhs();
      *rhs2 = fr.second_rhs();
    };
    hooks.prepare_pair = [&](double d0, double, double d1, double, const double** lhs2, const double** rhs2) {
      fr.build_pair(d0, d1, mu);
      *lhs2 = fr.second_lhs();
      *rhs2 = fr.second_rhs();
    };
    if (previous_took_first_attempt) hooks.after = [&](double d, double) { chain(d); };
    // (the eliminated rows' pivots without regularization: reduced with this iterate's error norms)
    hooks.eliminated_min_pivot = [&] { return cur.eliminated_min_pivot; };
    auto info = sys.compute_hooked(hooks);
    previous_took_first_attempt = sys.last_factorizations() == 1;
    if (info[0] == FactorInfo::Success && !sys.last_hooked_chain_valid()) chain(sys.hessian_regularization()[0]);
    if (info[0] == FactorInfo::Success) fr.wait_published();
    rep.factorizations += sys.last_factorizations();
    rep.solves += sys.last_factorizations();
    rep.value_sweeps += sys.last_factorizations();
    rep.t_kkt_decomp += since(t0);
    if (info[0] != FactorInfo::Success) return finish(ExitStatus::FACTORIZATION_FAILED);  // :463-465
    const double delta = sys.hessian_regularization()[0];
    rep.delta = delta;
    rep.gamma = sys.constraint_jacobian_regularization()[0];

    t0 = clk::now();
    const double alpha_max = H.dir.alpha_max;  // :488
    double alpha = alpha_max;
    double alpha_z = H.dir.alpha_z;  // :497
    const double D_phi = H.dir.D_phi;  // :508-509
    bool failed = alpha < alpha_min;
    const FilterEntry current_entry{cur.e.f - mu * cur.e.logsum, cur.e.viol};
    static const bool fr_debug = std::getenv("SLPX_FR_DEBUG") != nullptr;
    if (fr_debug) {
      std::fprintf(stderr, "fr: it %d mu %.3e delta %.1e gamma %.1e nfact %d | alpha_max %.3e alpha_z %.3e D_phi %.3e emin %.3e | cur f %.6e viol %.6e logsum %.6e | trial f %.6e viol %.6e logsum %.6e fin %g\n",
                   iterations, mu, delta, rep.gamma, sys.last_factorizations(), alpha_max, alpha_z, D_phi, H.dir.eliminated_min_pivot, cur.e.f, cur.e.viol,
                   cur.e.logsum, H.err_ahead.e.f, H.err_ahead.e.viol, H.err_ahead.e.logsum, H.err_ahead.e.finite);
    }
    double alpha_commit = alpha;
    bool have_trial = true;
    bool trial_is_ahead = true, took_lookahead = false;

    while (!failed) {  // :512
      if (!have_trial) {
        fr.trial_point(alpha);
        dev.sweep_values_trial();
        fr.trial_metrics(alpha, mu);
        fr.wait_published();
        ++rep.value_sweeps;
      }
      have_trial = false;
      alpha_commit = alpha;
      const bool from_ahead = trial_is_ahead;
      trial_is_ahead = false;
      IpmTrialOut tr = from_ahead ? IpmTrialOut{H.err_ahead.e.f, H.err_ahead.e.viol, H.err_ahead.e.logsum, H.err_ahead.e.finite} : H.trial;

      if (tr.finite == 0.0) {
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) failed = true;
        continue;
      }
      const FilterEntry trial_entry{tr.f - mu * tr.logsum, tr.viol};
      if (filter.try_add(current_entry, trial_entry, D_phi, alpha)) {
        took_lookahead = from_ahead;
        break;
      }

      const double prev_violation = cur.e.viol;
      double next_violation = tr.viol;
      // second-order corrections (:566-668): new right-hand side, SAME factorization
      if (alpha == alpha_max && next_violation >= prev_violation) {
        fr.save_direction();
        double alpha_soc = alpha, alpha_z_soc = alpha_z;
        double soc_violation = next_violation;
        bool step_acceptable = false;
        for (int it = 0; it < 5 && !step_acceptable; ++it) {
          fr.soc_accumulate(alpha_soc, it == 0);
          fr.build(delta, mu, /*soc=*/true, /*rhs_only=*/true);
          dev.solve();
          ++rep.solves;
          fr.expand(delta, mu, tau, /*soc=*/true);
          dev.sweep_values_trial();
          fr.trial_metrics(-1.0, mu);
          fr.wait_published();
          ++rep.value_sweeps;
          alpha_soc = H.dir.alpha_max;
          alpha_z_soc = H.dir.alpha_z;
          tr = H.trial;
          const FilterEntry soc_entry{tr.f - mu * tr.logsum, tr.viol};
          if (filter.try_add(current_entry, soc_entry, D_phi, alpha)) {
            alpha = alpha_soc;
            alpha_z = alpha_z_soc;
            alpha_commit = alpha_soc;
            step_acceptable = true;
            break;
          }
          next_violation = tr.viol;
          if (next_violation > 0.99 * soc_violation) break;
          soc_violation = next_violation;
        }
        if (step_acceptable) break;
        fr.restore_direction();
      }

      if (alpha == alpha_max) ++full_step_rejected_counter;
      // :677-684
      if (full_step_rejected_counter >= 4 && filter.max_constraint_violation > current_entry.constraint_violation / 10.0 &&
          filter.last_rejection_due_to_filter()) {
        filter.max_constraint_violation *= 0.1;
        filter.reset();
        have_trial = false;
        continue;
      }
      alpha *= alpha_reduction_factor;
      if (alpha < alpha_min) {  // :691-716 — rare: on the host, with the iterate pulled over
        host_current = false;
        pull_state();
        Vo.resize(st.nV);
        dev.download_V(Vo.data());
        const double current_kkt = restoration_kkt_error_one_norm(st, Vo, X, S, y, Z, x_r, w, rho, mu);
        Vec p(dim), ps0(m_i), pz0(m_i), dpn(M), psx(M), pzx(M);
        dev.download(dev.d_p(), p.data(), dim);
        if (m_i) {
          dev.download(dev.d_ps(), ps0.data(), m_i);
          dev.download(dev.d_pz(), pz0.data(), m_i);
        }
        fr.download_direction(dpn.data(), psx.data(), pzx.data());
        Vec Xt(X), St(S), yt(y), Zt(Z);
        for (int k = 0; k < n; ++k) Xt[k] += alpha_max * p[k];
        for (int e = 0; e < M; ++e) {
          Xt[n + e] += alpha_max * dpn[e];
          St[m_i + e] += alpha_max * psx[e];
          Zt[m_i + e] += alpha_z * pzx[e];
        }
        for (int r = 0; r < m_i; ++r) {
          St[r] += alpha_max * ps0[r];
          Zt[r] += alpha_z * pz0[r];
        }
        for (int j = 0; j < m_e; ++j) yt[j] += alpha_z * (-p[n + j]);
        // A_e, A_i at the trial point: a full sweep there, then the iterate back in place
        dev.upload_x(Xt.data());
        dev.upload_duals(St.data(), yt.data(), Zt.data());
        dev.sweep_full();
        Vec Vt(st.nV);
        dev.download_V(Vt.data());
        dev.upload_x(X.data());
        dev.upload_duals(S.data(), y.data(), Z.data());
        dev.sweep_full();
        const double next_kkt = restoration_kkt_error_one_norm(st, Vt, Xt, St, yt, Zt, x_r, w, rho, mu);
        if (next_kkt <= 0.999 * current_kkt) {
          alpha_commit = alpha_max;
          break;
        }
        failed = true;
      }
    }
    rep.t_line_search += since(t0);
    if (failed) return finish(ExitStatus::FEASIBILITY_RESTORATION_FAILED);  // :721-723

    t0 = clk::now();
    if (alpha == alpha_max) full_step_rejected_counter = 0;
    host_current = false;
    if (took_lookahead) {
      fr.accept_lookahead();  // :775-801 happened in the look-ahead launch; the buffers change roles
      cur = H.err_ahead;      // :809-812 and the norms: already there
    } else {
      fr.commit(alpha_commit, alpha_z, mu);  // :775-801
      // AD refresh (:809-812) and every norm the next decisions need
      dev.sweep_full();
      fr.errors(false, mu);
      fr.wait_published();
      cur = H.err;
    }
    rep.t_ad_refresh += since(t0);

    E_0 = E0_of(cur.e);
    if (E_0 > options.tolerance) {  // :819-832
      double E_mu = E_mu_of(cur.e, mu);
      while (mu > mu_min && E_mu <= 10.0 * mu) {
        update_barrier();
        E_mu = E_mu_of(cur.e, mu);
      }
    }
    if (options.diagnostics) {
      std::fprintf(stderr,
                   "%4d  err %.3e  f %.6e  |c| %.3e  mu %.1e  delta %.3e  gamma %.3e  alpha %.2e  alpha_z %.2e  nfact %d  (restoration)\n",
                   iterations, E_0, cur.e.f, cur.e.viol, mu, rep.delta, rep.gamma, alpha, alpha_z, sys.last_factorizations());
    }
    ++iterations;
    if (iterations >= options.max_iterations) return finish(ExitStatus::MAX_ITERATIONS_EXCEEDED);
    if (since(solve_start) > options.timeout) return finish(ExitStatus::TIMEOUT);
  }
  return finish(ExitStatus::SUCCESS);
}

// feasibility_restoration.hpp:347-628 (interior-point variant).  g = the outer problem's dense gradient at x.
ExitStatus feasibility_restoration(NewtonSystem& outer, const Vec& scales, const std::vector<IterationCallback>& user_callbacks,
                                   const std::function<bool(const FilterEntry&, double)>& outer_accepts, const Options& options,
                                   Vec& x, Vec& s, Vec& y, Vec& z, double mu, int& iterations, SolveReport& rep,
                                   clk::time_point solve_start, const Vec& c_e, const Vec& c_i, const Vec& g, double initial_violation) {
  const NlpStructure& ost = outer.structure();
  const int n = ost.n, m_e = ost.m_e, m_i = ost.m_i, M = 2 * m_e + 2 * m_i;
  constexpr double rho = 1e3;
  ++rep.restorations;

  Vec cis(m_i);
  for (int j = 0; j < m_i; ++j) cis[j] = c_i[j] - s[j];
  const double fr_mu = std::max({mu, norm_inf(c_e.data(), m_e), norm_inf(cis.data(), m_i)});  // :396-397
  const double zeta = std::sqrt(fr_mu);

  Vec p_e, n_e, p_i, n_i;
  compute_p_n(c_e, rho, fr_mu, p_e, n_e);  // :400-401
  compute_p_n(cis, rho, fr_mu, p_i, n_i);

  DeviceNlp& dev = outer.device();
  const auto t_build = clk::now();
  dev.ipm_enable();  // (the trial buffers)
  FrDevice& fr = outer.restoration_device();
  rep.t_restoration_setup += since(t_build);

  // the restoration iterate (:408-423): X = [x | p_e | n_e | p_i | n_i], S = [s | 1...], y = 0, Z = fr_mu / S (.. / p, n)
  Vec X;
  X.reserve(n + M);
  for (const Vec* v : {static_cast<const Vec*>(&x), static_cast<const Vec*>(&p_e), static_cast<const Vec*>(&n_e),
                       static_cast<const Vec*>(&p_i), static_cast<const Vec*>(&n_i)})
    X.insert(X.end(), v->begin(), v->end());
  Vec S(m_i + M, 1.0), fr_y(m_e, 0.0), Z;
  std::copy(s.begin(), s.end(), S.begin());
  Z.reserve(m_i + M);
  for (int j = 0; j < m_i; ++j) Z.push_back(fr_mu * (1.0 / s[j]));
  for (int e = 0; e < M; ++e) Z.push_back(fr_mu * (1.0 / X[n + e]));
  const Vec x_r(x);
  Vec w(n);
  for (int k = 0; k < n; ++k) w[k] = zeta * std::min(1.0 / (x[k] * x[k]), 1.0);  // zeta D_R (:404-405)

  // the error measure un-scales with {1, d_ce, [d_ci, 1...]} (:425-432); the bound rows' 1 is implied
  Vec fr_scales(1 + m_e + m_i, 1.0);
  for (int j = 0; j < m_e; ++j) fr_scales[1 + j] = scales[1 + j];
  for (int j = 0; j < m_i; ++j) fr_scales[1 + m_e + j] = scales[1 + m_e + j];

  const Vec s_outer(s);
  fr.begin(x_r.data(), w.data(), g.data(), s_outer.data(), mu, X.data() + n, S.data() + m_i, Z.data() + m_i, fr_scales);
  dev.upload_x(X.data());
  dev.upload_duals(S.data(), fr_y.data(), Z.data());

  // the callback the reference appends (interior_point.hpp:729-752): leave once the OUTER filter accepts the
  // restoration iterate and the violation dropped by 10 %
  auto accept_exit = [&](const FrErrOut& e) {
    const FilterEntry trial_entry{e.f_outer - mu * e.logsum_outer, e.viol_outer};
    return trial_entry.constraint_violation < 0.9 * initial_violation && outer_accepts(trial_entry, e.dphi_outer);
  };

  const auto saved_regularization = outer.regularization_state();
  double mu_fr = fr_mu;
  const int it_before = iterations;
  const auto t_inner = clk::now();
  const double outer_error = rep.final_error;
  const ExitStatus status = restoration_core(outer, fr, user_callbacks, accept_exit, options, x_r, w, X, S, fr_y, Z, mu_fr, iterations,
                                             rep, solve_start);
  outer.set_regularization_state(saved_regularization);
  outer.set_gamma_min(1e-10);
  rep.final_error = outer_error;
  rep.restoration_iterations += iterations - it_before;
  rep.t_restoration += since(t_inner);

  std::copy(X.begin(), X.begin() + n, x.begin());
  std::copy(S.begin(), S.begin() + m_i, s.begin());

  if (status == ExitStatus::CALLBACK_REQUESTED_STOP) {
    // back to the original problem: least-squares multipliers at the new point (:606-617)
    Vec V(ost.nV);
    dev.upload_x(x.data());
    dev.upload_duals(s.data(), y.data(), z.data());
    dev.sweep_full();
    dev.download_V(V.data());
    if (!lagrange_multiplier_estimate(outer, V, s, mu, y, z, rep)) return ExitStatus::FACTORIZATION_FAILED;
    return ExitStatus::SUCCESS;
  } else if (status == ExitStatus::SUCCESS) {
    return ExitStatus::LOCALLY_INFEASIBLE;
  }
  return ExitStatus::FEASIBILITY_RESTORATION_FAILED;
}

}  // namespace

// ---------------------------------------------------------------------------
// The reference's other two solvers on the same device Newton step:
// Problem::solve sends problems without constraints to newton() and problems with equality
// constraints only to sqp() (problem.hpp:335, 403).  Both are the interior-point iteration
// with its inequality machinery taken out — but not quite the m_i = 0 special case of it: SQP
// moves y with the primal step size (the IPM would take alpha_z = 1), Newton's line search
// goes down to alpha = 1e-20 and ends in LINE_SEARCH_FAILED instead of a restoration phase.
// Host-resident drivers (one value sweep per trial point crosses PCIe): these are the small
// problems of the reference's unit tests, not the benchmark path.
// ---------------------------------------------------------------------------
namespace {

// sqp.hpp:98-604
ExitStatus sqp_core(NewtonSystem& sys, const Vec& scales, const std::vector<IterationCallback>& callbacks,
                    const Options& options, Vec& x, Vec& y, int& iterations, SolveReport& rep,
                    clk::time_point solve_start) {
  const NlpStructure& st = sys.structure();
  DeviceNlp& dev = sys.device();
  const int n = st.n, m_e = st.m_e, dim = n + m_e;
  const Vec none;  // no slacks, no inequality multipliers
  double mu0 = 0.0;

  sys.reset_regularization();
  sys.set_gamma_min(1e-10);  // sparse_regularized_ldlt.hpp:197 (the constructor of sqp.hpp:238-240 passes none)

  Vec V(st.nV), Vtrial(st.nV);
  auto refresh_full = [&](const Vec& xx, const Vec& yy) {
    dev.upload_x(xx.data());
    dev.upload_duals(none.data(), yy.data(), none.data());
    dev.sweep_full();
    dev.download_V(V.data());
  };
  auto eval_values = [&](const Vec& xx) {
    dev.upload_x(xx.data());
    dev.sweep_values();
    dev.download(dev.d_V(), Vtrial.data(), static_cast<size_t>(st.off_g));
    ++rep.value_sweeps;
  };

  auto t_setup = clk::now();
  refresh_full(x, y);  // :182-186
  VView cur{st, V};
  Vec g = cur.g_dense();
  if (m_e > n) return ExitStatus::TOO_FEW_DOFS;                                   // :205-210
  if (!all_finite(V.data(), st.nV)) return ExitStatus::NONFINITE_INITIAL_GUESS;  // :213-216

  double f = cur.f();
  Vec c_e(cur.c_e(), cur.c_e() + m_e);
  Filter filter{norm_1(c_e.data(), m_e)};  // :220
  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-7;
  int full_step_rejected_counter = 0;
  const bool identity = scaling_is_identity(st, scales);
  auto E0_of = [&](const Vec& gg, const Vec& ce, const Vec& yy) {
    return kkt_error_impl<ErrType::INF_NORM_SCALED>(st, gg, cur.Ae(), ce.data(), nullptr, nullptr, none, yy, none,
                                                    0.0, identity ? nullptr : &scales);
  };
  double E_0 = E0_of(g, c_e, y);  // :253-254
  rep.t_setup = since(t_setup);

  Vec p(dim), p_x(n), p_y(m_e), trial_x, trial_y, trial_c_e(m_e);
  double trial_f = 0.0;
  auto read_trial = [&] {
    trial_f = Vtrial[st.off_f];
    std::copy(Vtrial.begin() + st.off_ce, Vtrial.begin() + st.off_ce + m_e, trial_c_e.begin());
  };
  auto solve_step = [&](Vec& px, Vec& py) {
    dev.download(dev.d_p(), p.data(), dim);
    std::copy(p.begin(), p.begin() + n, px.begin());
    for (int j = 0; j < m_e; ++j) py[j] = -p[n + j];
  };

  while (E_0 > options.tolerance) {
    // :277-292 infeasibility / divergence checks
    if (m_e > 0) {
      Vec t(n, 0.0);
      add_At_v(st.Ae, cur.Ae(), nullptr, c_e.data(), 1.0, t);
      double nt = 0.0, nc = 0.0;
      for (double v : t) nt += v * v;
      for (double v : c_e) nc += v * v;
      if (std::sqrt(nt) < 1e-6 && std::sqrt(nc) > 1e-2) return ExitStatus::LOCALLY_INFEASIBLE;
    }
    if (norm_inf(x.data(), n) > 1e10 || !all_finite(x.data(), n)) return ExitStatus::DIVERGING_ITERATES;
    for (const auto& cb : callbacks)
      if (cb({iterations, x, none, y, none, V, &st, false})) return ExitStatus::CALLBACK_REQUESTED_STOP;

    // ---- Newton-KKT step on the device: [H A_e^T; A_e 0] [p_x; -p_y] = -[g - A_e^T y; c_e] (:305-346) ----
    auto t0 = clk::now();
    dev.upload_x(x.data());
    dev.upload_duals(none.data(), y.data(), none.data());
    dev.upload_mu(&mu0);
    dev.assemble();
    dev.build_rhs();
    rep.t_kkt_build += since(t0);
    t0 = clk::now();
    auto info = sys.compute(/*solve_speculatively=*/true);
    rep.factorizations += sys.last_factorizations();
    rep.t_kkt_decomp += since(t0);
    if (info[0] != FactorInfo::Success) return ExitStatus::FACTORIZATION_FAILED;  // :336-338
    rep.delta = sys.hessian_regularization()[0];
    rep.gamma = sys.constraint_jacobian_regularization()[0];
    t0 = clk::now();
    rep.solves += sys.last_factorizations();
    solve_step(p_x, p_y);
    rep.t_kkt_solve += since(t0);

    t0 = clk::now();
    constexpr double alpha_max = 1.0;
    double alpha = alpha_max;
    bool call_feasibility_restoration = false;
    const FilterEntry current_entry{f, norm_1(c_e.data(), m_e)};
    double D_phi = 0.0;  // :360
    for (int i = 0; i < n; ++i) D_phi += g[i] * p_x[i];

    while (true) {  // :364
      trial_x = axpy(x, alpha, p_x);
      trial_y = axpy(y, alpha, p_y);
      eval_values(trial_x);
      read_trial();
      if (!std::isfinite(trial_f) || !all_finite(trial_c_e.data(), m_e)) {  // :373-384
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) {
          call_feasibility_restoration = true;
          break;
        }
        continue;
      }
      if (filter.try_add(current_entry, FilterEntry{trial_f, norm_1(trial_c_e.data(), m_e)}, D_phi, alpha)) break;

      const double prev_violation = norm_1(c_e.data(), m_e);
      double next_violation = norm_1(trial_c_e.data(), m_e);
      // second-order corrections (:397-468): new bottom rows of the rhs, SAME factorization
      if (alpha == alpha_max && next_violation >= prev_violation) {
        Vec soc_px = p_x, soc_py = p_y, c_e_soc = c_e;
        const double alpha_soc = alpha;
        double soc_violation = next_violation;
        bool step_acceptable = false;
        for (int it = 0; it < 5 && !step_acceptable; ++it) {
          for (int j = 0; j < m_e; ++j) c_e_soc[j] = alpha_soc * c_e_soc[j] + trial_c_e[j];  // :435
          Vec rhs(dim, 0.0);
          for (int i = 0; i < n; ++i) rhs[i] = -g[i];
          add_At_v(st.Ae, cur.Ae(), nullptr, y.data(), 1.0, rhs);
          for (int j = 0; j < m_e; ++j) rhs[n + j] = -c_e_soc[j];
          SLPX_HIP_CHECK(hipMemcpyAsync(dev.d_rhs(), rhs.data(), dim * sizeof(double), hipMemcpyHostToDevice,
                                        dev.stream()));
          dev.solve();
          ++rep.solves;
          solve_step(soc_px, soc_py);
          trial_x = axpy(x, alpha_soc, soc_px);
          trial_y = axpy(y, alpha_soc, soc_py);
          eval_values(trial_x);
          read_trial();
          if (filter.try_add(current_entry, FilterEntry{trial_f, norm_1(trial_c_e.data(), m_e)}, D_phi, alpha)) {
            p_x = soc_px;
            p_y = soc_py;
            alpha = alpha_soc;
            step_acceptable = true;
            break;
          }
          constexpr double kappa_soc = 0.99;
          next_violation = norm_1(trial_c_e.data(), m_e);
          if (next_violation > kappa_soc * soc_violation) break;
          soc_violation = next_violation;
        }
        if (step_acceptable) break;
      }
      if (alpha == alpha_max) ++full_step_rejected_counter;  // :472-474
      if (full_step_rejected_counter >= 4 && filter.max_constraint_violation > current_entry.constraint_violation / 10.0 &&
          filter.last_rejection_due_to_filter()) {  // :478-485
        filter.max_constraint_violation *= 0.1;
        filter.reset();
        continue;
      }
      alpha *= alpha_reduction_factor;
      if (alpha < alpha_min) {  // :492-517
        const double current_kkt = kkt_error_impl<ErrType::ONE_NORM>(st, g, cur.Ae(), c_e.data(), nullptr, nullptr, none,
                                                                     y, none, 0.0, nullptr);
        trial_x = axpy(x, alpha_max, p_x);
        trial_y = axpy(y, alpha_max, p_y);
        Vec Vt(st.nV);
        dev.upload_x(trial_x.data());
        dev.upload_duals(none.data(), trial_y.data(), none.data());
        dev.sweep_full();
        dev.download_V(Vt.data());
        VView tv{st, Vt};
        trial_f = tv.f();
        std::copy(tv.c_e(), tv.c_e() + m_e, trial_c_e.begin());
        const double next_kkt = kkt_error_impl<ErrType::ONE_NORM>(st, tv.g_dense(), tv.Ae(), tv.c_e(), nullptr, nullptr,
                                                                  none, trial_y, none, 0.0, nullptr);
        if (next_kkt <= 0.999 * current_kkt) break;
        call_feasibility_restoration = true;
        break;
      }
    }
    rep.t_line_search += since(t0);

    if (call_feasibility_restoration) {  // :521-556
      refresh_full(x, y);  // the device V describes x again (the fallback above moved it)
      const double f_x = cur.f();
      const FilterEntry initial_entry{f_x, norm_1(c_e.data(), m_e)};
      const Vec g_here = cur.g_dense();
      auto outer_accepts = [&, alpha](const FilterEntry& trial_entry, double D_phi_restoration) {
        return filter.try_add(initial_entry, trial_entry, D_phi_restoration, alpha);
      };
      // feasibility_restoration.hpp:103-345: the interior-point variant with no inequality rows
      // of the original problem and mu = tolerance / 10
      Vec s_none, z_none;
      const ExitStatus fr_status = feasibility_restoration(sys, scales, callbacks, outer_accepts, options, x, s_none, y, z_none,
                                                           options.tolerance / 10.0, iterations, rep, solve_start, c_e, Vec{}, g_here,
                                                           initial_entry.constraint_violation);
      if (fr_status != ExitStatus::SUCCESS) return fr_status;
      sys.set_gamma_min(1e-10);
    } else {
      if (alpha == 1.0) full_step_rejected_counter = 0;
      x = trial_x;
      y = trial_y;
    }
    auto t1 = clk::now();
    refresh_full(x, y);  // :574-577
    rep.t_ad_refresh += since(t1);
    g = cur.g_dense();
    f = cur.f();
    std::copy(cur.c_e(), cur.c_e() + m_e, c_e.begin());
    E_0 = E0_of(g, c_e, y);
    rep.final_error = E_0;
    if (options.diagnostics)
      std::fprintf(stderr, "%4d  err %.3e  f %.6e  |c| %.3e  mu %.1e  delta %.3e  gamma %.3e  alpha %.2e  alpha_z %.2e  nfact %d  (sqp)\n",
                   iterations, E_0, f, norm_1(c_e.data(), m_e), 0.0, rep.delta, rep.gamma, alpha, alpha, sys.last_factorizations());
    ++iterations;
    if (iterations >= options.max_iterations) return ExitStatus::MAX_ITERATIONS_EXCEEDED;
    if (since(solve_start) > options.timeout) return ExitStatus::TIMEOUT;
  }
  return ExitStatus::SUCCESS;
}

// newton.hpp:51-292
ExitStatus newton_core(NewtonSystem& sys, const Vec& scales, const std::vector<IterationCallback>& callbacks,
                       const Options& options, Vec& x, int& iterations, SolveReport& rep,
                       clk::time_point solve_start) {
  const NlpStructure& st = sys.structure();
  DeviceNlp& dev = sys.device();
  const int n = st.n;
  const Vec none;
  double mu0 = 0.0;
  sys.reset_regularization();
  sys.set_gamma_min(1e-10);  // no equality rows: unused

  Vec V(st.nV), Vtrial(st.nV);
  auto refresh_full = [&](const Vec& xx) {
    dev.upload_x(xx.data());
    dev.sweep_full();
    dev.download_V(V.data());
  };
  auto eval_f = [&](const Vec& xx) {
    dev.upload_x(xx.data());
    dev.sweep_values();
    dev.download(dev.d_V(), Vtrial.data(), static_cast<size_t>(st.off_g));
    ++rep.value_sweeps;
    return Vtrial[st.off_f];
  };
  refresh_full(x);
  VView cur{st, V};
  Vec g = cur.g_dense();
  if (!all_finite(V.data(), st.nV)) return ExitStatus::NONFINITE_INITIAL_GUESS;  // :125-127
  double f = cur.f();
  Filter filter{0.0};  // :131
  constexpr double alpha_reduction_factor = 0.5, alpha_min = 1e-20;  // :138-139
  const bool identity = scaling_is_identity(st, scales);
  auto E0_of = [&](const Vec& gg) {
    return kkt_error_impl<ErrType::INF_NORM_SCALED>(st, gg, nullptr, nullptr, nullptr, nullptr, none, none, none, 0.0,
                                                    identity ? nullptr : &scales);
  };
  double E_0 = E0_of(g);
  Vec p_x(n), trial_x;
  double trial_f = 0.0;
  while (E_0 > options.tolerance) {
    if (norm_inf(x.data(), n) > 1e10 || !all_finite(x.data(), n)) return ExitStatus::DIVERGING_ITERATES;  // :164
    for (const auto& cb : callbacks)
      if (cb({iterations, x, none, none, none, V, &st, false})) return ExitStatus::CALLBACK_REQUESTED_STOP;
    // H p_x = -g (:182-190)
    auto t0 = clk::now();
    dev.upload_x(x.data());
    dev.upload_mu(&mu0);
    dev.assemble();
    dev.build_rhs();
    auto info = sys.compute(/*solve_speculatively=*/true);
    rep.factorizations += sys.last_factorizations();
    rep.solves += sys.last_factorizations();
    rep.t_kkt_decomp += since(t0);
    if (info[0] != FactorInfo::Success) return ExitStatus::FACTORIZATION_FAILED;
    rep.delta = sys.hessian_regularization()[0];
    dev.download(dev.d_p(), p_x.data(), n);

    t0 = clk::now();
    constexpr double alpha_max = 1.0;
    double alpha = alpha_max;
    double D_phi = 0.0;
    for (int i = 0; i < n; ++i) D_phi += g[i] * p_x[i];
    while (true) {  // :201-243
      trial_x = axpy(x, alpha, p_x);
      trial_f = eval_f(trial_x);
      if (!std::isfinite(trial_f)) {
        alpha *= alpha_reduction_factor;
        if (alpha < alpha_min) return ExitStatus::LINE_SEARCH_FAILED;
        continue;
      }
      if (filter.try_add(FilterEntry{f, 0.0}, FilterEntry{trial_f, 0.0}, D_phi, alpha)) break;
      alpha *= alpha_reduction_factor;
      if (alpha < alpha_min) {
        const double current_kkt = norm_1(g.data(), n);
        trial_x = axpy(x, alpha_max, p_x);
        Vec Vt(st.nV);
        dev.upload_x(trial_x.data());
        dev.sweep_full();
        dev.download_V(Vt.data());
        VView tv{st, Vt};
        const Vec tg = tv.g_dense();
        if (norm_1(tg.data(), n) <= 0.999 * current_kkt) {
          trial_f = tv.f();
          break;
        }
        return ExitStatus::LINE_SEARCH_FAILED;
      }
    }
    rep.t_line_search += since(t0);
    x = trial_x;
    f = trial_f;
    refresh_full(x);  // :254-255
    g = cur.g_dense();
    E_0 = E0_of(g);
    rep.final_error = E_0;
    if (options.diagnostics)
      std::fprintf(stderr, "%4d  err %.3e  f %.6e  |c| %.3e  mu %.1e  delta %.3e  gamma %.3e  alpha %.2e  alpha_z %.2e  nfact %d  (newton)\n",
                   iterations, E_0, f, 0.0, 0.0, rep.delta, 0.0, alpha, alpha, sys.last_factorizations());
    ++iterations;
    if (iterations >= options.max_iterations) return ExitStatus::MAX_ITERATIONS_EXCEEDED;
    if (since(solve_start) > options.timeout) return ExitStatus::TIMEOUT;
  }
  return ExitStatus::SUCCESS;
}

}  // namespace

ExitStatus sqp(NewtonSystem& sys, const std::vector<double>& scales, const std::vector<IterationCallback>& callbacks,
               const Options& options, std::vector<double>& x, std::vector<double>* y_out, SolveReport* report) {
  const auto solve_start = clk::now();
  SolveReport local;
  SolveReport& rep = report ? *report : local;
  rep = SolveReport{};
  Vec y(sys.structure().m_e, 0.0);  // problem.hpp:503-504
  int iterations = 0;
  const ExitStatus status = sqp_core(sys, scales, callbacks, options, x, y, iterations, rep, solve_start);
  if (y_out) *y_out = y;
  rep.iterations = iterations;
  rep.t_total = since(solve_start);
  return status;
}

ExitStatus newton(NewtonSystem& sys, const std::vector<double>& scales, const std::vector<IterationCallback>& callbacks,
                  const Options& options, std::vector<double>& x, SolveReport* report) {
  const auto solve_start = clk::now();
  SolveReport local;
  SolveReport& rep = report ? *report : local;
  rep = SolveReport{};
  int iterations = 0;
  const ExitStatus status = newton_core(sys, scales, callbacks, options, x, iterations, rep, solve_start);
  rep.iterations = iterations;
  rep.t_total = since(solve_start);
  return status;
}

ExitStatus feasibility_restoration_steps(NewtonSystem& sys, const std::vector<double>& scales,
                                         const Options& options, std::vector<double>& x,
                                         std::vector<double>& s, std::vector<double>& y,
                                         std::vector<double>& z, double mu, int steps,
                                         SolveReport* report) {
  const NlpStructure& st = sys.structure();
  SolveReport local;
  SolveReport& rep = report ? *report : local;
  rep = SolveReport{};
  // c_e, c_i at x (the caller of feasibility_restoration holds them, interior_point.hpp:721-727)
  DeviceNlp& dev = sys.device();
  Vec V(st.nV);
  dev.upload_x(x.data());
  dev.upload_duals(s.data(), y.data(), z.data());
  dev.sweep_full();
  dev.download_V(V.data());
  const Vec c_e(V.begin() + st.off_ce, V.begin() + st.off_ce + st.m_e);
  const Vec c_i(V.begin() + st.off_ci, V.begin() + st.off_ci + st.m_i);
  int iterations = 0;
  const std::vector<IterationCallback> stop{
      [steps](const IterationInfo& info) { return info.iteration >= steps; }};
  const Vec g = VView{st, V}.g_dense();
  double violation = norm_1(c_e.data(), st.m_e);
  for (int j = 0; j < st.m_i; ++j) violation += std::abs(c_i[j] - s[j]);
  const auto never = [](const FilterEntry&, double) { return false; };
  return feasibility_restoration(sys, scales, stop, never, options, x, s, y, z, mu, iterations, rep, clk::now(), c_e, c_i, g,
                                 violation);
}

ExitStatus interior_point(NewtonSystem& sys, const std::vector<double>& scales,
                          const std::vector<IterationCallback>& callbacks, const Options& options,
                          std::vector<double>& x, std::vector<double>* s_out,
                          std::vector<double>* y_out, std::vector<double>* z_out,
                          SolveReport* report) {
  const NlpStructure& st = sys.structure();
  const auto solve_start = clk::now();
  SolveReport local;
  SolveReport& rep = report ? *report : local;
  rep = SolveReport{};
  // interior_point.hpp:74-79
  Vec s(st.m_i, 1.0), y(st.m_e, 0.0), z(st.m_i, 1.0);
  double mu = 0.1 * scales[0];
  int iterations = 0;
  const ExitStatus status = ipm_core(sys, scales, callbacks, options, false, x, s, y, z, mu, iterations, rep, solve_start);
  rep.t_total = since(solve_start);
  if (s_out) *s_out = s;
  if (y_out) *y_out = y;
  if (z_out) *z_out = z;
  rep.iterations = iterations;
  return status;
}

}  // namespace slpx
