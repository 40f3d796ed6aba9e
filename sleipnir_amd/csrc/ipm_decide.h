// The DECISIONS of the interior-point iteration as plain functions of a few dozen scalars, shared by the host driver
// (ipm.cpp) and by the launch that makes them on the device for the common iteration (ipm_kernels.h: the fold of the
// look-ahead error norms): the filter (util/filter.hpp:17-212), the error measures (util/kkt_error.hpp:92-146 from the
// reduced scalars), and the test "this iteration took its full step and the next one needs nothing from the host".
// One source for both sides, so that what the device decides is what the host would have decided.
#pragma once

#include <cmath>

#include "device.hpp"

#ifdef __HIPCC__
#define SLPX_DECIDE __host__ __device__ inline
#else
#define SLPX_DECIDE inline
#endif

namespace slpx {

// filter.hpp:17-60
struct FilterEntry {
  double cost = 0.0, constraint_violation = 0.0;
};
SLPX_DECIDE bool ipm_isfinite(double x) { return x - x == 0.0; }
SLPX_DECIDE bool filter_dominated_by(const FilterEntry& a, const FilterEntry& e) {
  return e.cost <= a.cost && e.constraint_violation <= a.constraint_violation;
}

// filter.hpp:62-212 as a value: the entries in the order they were added, (cost, violation) pairs.  (A full table turns
// the device's decisions off, never the filter: the host driver keeps a vector of its own, Filter in ipm.cpp, and
// applies the same filter_rules() to it.)
constexpr int kFilterCapacity = 1024;
struct FilterState {
  double min_constraint_violation = 0.0, max_constraint_violation = 0.0;
  int n = 0;
  int last_rejection_due_to_filter = 0;
  double ent[2 * kFilterCapacity];  // LAST member: only the first 2 n doubles travel (DeviceNlp::ipm_pipeline_upload)
};

constexpr double kFilterGammaCost = 1e-8, kFilterGammaCon = 1e-5;

// The three powers of the filter's rules (filter.hpp:118-131).  The host takes them one after the other; the device
// deals them to three lanes of the deciding wave — one pass through pow() instead of three (ipm_kernels.h).
struct FilterPowers {
  double dphi_23 = 0.0;   // (-D_phi)^2.3   (used only where D_phi < 0)
  double viol_11 = 0.0;   // (current constraint violation)^1.1
  double alpha_15 = 0.0;  // alpha^1.5
};
SLPX_DECIDE FilterPowers filter_powers(const FilterEntry& cur, double D_phi, double alpha) {
#pragma clang fp contract(off)  // (host and device round alike: no fused multiply-adds)
  return FilterPowers{pow(-D_phi, 2.3), pow(cur.constraint_violation, 1.1), pow(alpha, 1.5)};
}

// filter.hpp:109-172 up to the table itself: 0 rejected (switching / Armijo / sufficient-decrease rules; the rejection
// flag cleared as the reference does), 1 goes on to the table.  `add`: the entry an acceptance would insert, if *insert.
SLPX_DECIDE int filter_rules(double min_constraint_violation, double max_constraint_violation, int* last_rejection_due_to_filter,
                             const FilterEntry& cur, const FilterEntry& trial, double D_phi, double alpha, const FilterPowers& pw,
                             FilterEntry* add, bool* insert) {
#pragma clang fp contract(off)  // (host and device round alike: no fused multiply-adds)
  if (!ipm_isfinite(trial.cost) || trial.constraint_violation > max_constraint_violation) return 0;
  const bool switching = D_phi < 0.0 && alpha * pw.dphi_23 > pw.viol_11;
  const bool armijo = trial.cost <= cur.cost + 1e-8 * alpha * D_phi;
  const double phi = pw.alpha_15;
  const bool sufficient = trial.cost <= cur.cost - phi * kFilterGammaCost * cur.constraint_violation ||
                          trial.constraint_violation <= (1.0 - phi * kFilterGammaCon) * cur.constraint_violation;
  if (cur.constraint_violation <= min_constraint_violation && switching) {
    if (!armijo) {
      *last_rejection_due_to_filter = 0;
      return 0;
    }
  } else if (!sufficient) {
    *last_rejection_due_to_filter = 0;
    return 0;
  }
  *insert = !switching || !armijo;
  *add = FilterEntry{cur.cost - phi * kFilterGammaCost * cur.constraint_violation, (1.0 - phi * kFilterGammaCon) * cur.constraint_violation};
  return 1;
}

// util/kkt_error.hpp:92-146 from the reduced scalars (IpmErrOut, device.hpp)
SLPX_DECIDE double ipm_max4(double a, double b, double c, double d) { return fmax(fmax(a, b), fmax(c, d)); }
SLPX_DECIDE double ipm_E_mu(const IpmErrOut& e, double m, int m_e, int m_i) {
#pragma clang fp contract(off)  // (host and device round alike: no fused multiply-adds)
  constexpr double s_max = 100.0;
  const double s_d = fmax(s_max, (e.y1 + e.z1) / double(m_e + m_i)) / s_max;
  const double s_c = fmax(s_max, e.z1 / double(m_i)) / s_max;
  const double comp = m_i ? fmax(fabs(e.sz_max - m), fabs(e.sz_min - m)) : 0.0;
  return ipm_max4(e.dual_inf / s_d, comp / s_c, e.ce_inf, e.cis_inf);
}
SLPX_DECIDE double ipm_E_0(const IpmErrOut& e, int m_e, int m_i, bool identity_scaling) {
#pragma clang fp contract(off)  // (host and device round alike: no fused multiply-adds)
  if (identity_scaling) return ipm_E_mu(e, 0.0, m_e, m_i);
  constexpr double s_max = 100.0;
  const double s_d = fmax(s_max, (e.y1_u + e.z1_u) / double(m_e + m_i)) / s_max;
  const double s_c = fmax(s_max, e.z1_u / double(m_i)) / s_max;
  return ipm_max4(e.dual_inf_u / s_d, e.sz_max_u / s_c, e.ce_inf_u, e.cis_inf_u);
}

// What the device keeps of the iteration's state to decide the common case itself (DeviceNlp::ipm_pipeline_*): the host
// uploads it whenever it took a decision of its own, and fetches the filter back when it has to take one again.
struct IpmCtl {
  double mu = 0.0, mu_min = 0.0, tolerance = 0.0;
  double cur_f = 0.0, cur_logsum = 0.0, cur_viol = 0.0;  // the current iterate's filter quantities
  int m_e = 0, m_i = 0, identity_scaling = 0;
  int go = 0;  // (the last decision)
  FilterState filter;  // LAST (its used prefix travels)
};

// interior_point.hpp:387-408 and :809-832 on the scalars of the iterate the look-ahead launch evaluated — everything
// of the common iteration's decision but the filter: true = the error is above the tolerance, the barrier parameter
// stays, nothing is infeasible or diverging, the step has a size.
SLPX_DECIDE bool ipm_next_iteration_is_plain(const IpmCtl& C, const IpmErrOut& ahead, double alpha_max) {
#pragma clang fp contract(off)  // (host and device round alike: no fused multiply-adds)
  constexpr double alpha_min = 1e-7;
  if (!(alpha_max >= alpha_min) || ahead.finite == 0.0) return false;
  if (!(ipm_E_0(ahead, C.m_e, C.m_i, C.identity_scaling != 0) > C.tolerance)) return false;
  if (C.mu > C.mu_min && ipm_E_mu(ahead, C.mu, C.m_e, C.m_i) <= 10.0 * C.mu) return false;  // (a barrier update: the host's)
  if (C.m_e > 0 && sqrt(ahead.aetce_sq) < 1e-6 && sqrt(ahead.ce_sq) > 1e-2) return false;
  if (C.m_i > 0 && sqrt(ahead.aitcp_sq) < 1e-6 && sqrt(ahead.cp_sq) > 1e-6) return false;
  if (ahead.x_inf > 1e10 || ahead.s_inf > 1e10) return false;
  return true;
}

}  // namespace slpx
