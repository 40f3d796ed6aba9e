// Feasibility restoration (include/sleipnir/optimization/solver/util/feasibility_restoration.hpp:347-628) ON THE OUTER
// PROBLEM'S OWN SYSTEM: no second model, no second compile, no larger factorization.
//
// The reference composes the restoration problem
//
//     min  rho sum(p_e + n_e + p_i + n_i) + zeta/2 (x - x_R)^T D_R (x - x_R)
//     s.t. c_e(x) - p_e + n_e  = 0
//          c_i(x) - p_i + n_i >= 0,   p_e, n_e, p_i, n_i >= 0
//
// out of the outer problem's callbacks (:434-593) and runs interior_point() on it (:602): n + 2 m_e + 2 m_i
// variables, m_i + 2 m_e + 2 m_i inequality rows, a KKT system of order n + 3 m_e + 2 m_i.  The extra variables
// enter with identity Jacobian columns ([A_e -I I 0 0], [A_i 0 0 -I I; 0 I ...], :499-573) and no Hessian, so their
// rows of the regularized Newton-KKT system (interior_point.hpp:426-448) are solved for in closed form, row by row:
// with Sigma_k = z_k / s_k of the five inequality blocks (0: c_i - p_i + n_i, 1: p_e, 2: n_e, 3: p_i, 4: n_i), the
// regularization (delta, gamma) of sparse_regularized_ldlt.hpp:82-151 and w = -p_y,
//
//     dp_e = (r_pe + w) / (Sigma_1 + delta),           dn_e = (r_ne - w) / (Sigma_2 + delta),
//     [Sigma_0+Sigma_3+delta  -Sigma_0 ] [dp_i]   [r_pi + Sigma_0 A_i dx]
//     [-Sigma_0  Sigma_0+Sigma_4+delta ] [dn_i] = [r_ni - Sigma_0 A_i dx]
//
// and what is left is a system with THE OUTER PROBLEM'S PATTERN:
//
//     [zeta D_R + H_c + delta I + A_i^T Sigma_eff A_i    A_e^T                                    ] [dx]   [r_x + A_i^T r_adj]
//     [A_e          -(gamma + 1/(Sigma_1+delta) + 1/(Sigma_2+delta))                              ] [w ] = [r_y + r_pe/(Sigma_1+delta) - r_ne/(Sigma_2+delta)]
//
//     1/Sigma_eff = 1/Sigma_0 + 1/(Sigma_3+delta) + 1/(Sigma_4+delta)
//
// — the outer system's KKT plan, symbolic factorization and step kernel factor and solve it (the eliminated block is
// positive definite, so by inertia additivity the big system has the inertia (n', m_e, 0) exactly when this one has
// (n, m_e, 0); the eliminated pivots join the |D| >= 1e-4 test of the unregularized attempt, :82-87).  Everything else of
// the restoration iteration — the full direction, step sizes, trial points, filter quantities, second-order corrections,
// the iterate update, the error norms, and the OUTER problem's filter quantities that decide when restoration ends
// (interior_point.hpp:729-752) — is a handful of row-local kernels on the restoration iterate, whose x, s_0, y, z_0 live
// in the outer system's own buffers (the outer tape, swept at x with (y, z_0), IS the restoration problem's c_e, c_i,
// A_e, A_i, H_c: :434-593 call the outer callbacks) and whose p, n and their slacks / duals live here.
#pragma once

#include <memory>
#include <vector>

#include "device.hpp"

namespace slpx {

// what the restoration kernels hand the host (pinned memory)
struct FrDirOut {
  double alpha_max, alpha_z, D_phi;
  double eliminated_min_pivot;  // min of the pivots of the eliminated rows (order p_e, n_e, p_i, n_i)
};
struct FrErrOut {
  IpmErrOut e;  // the restoration problem's error norms (kkt_error.hpp) — e.f, e.viol, e.logsum are its filter entry
  // the OUTER problem at (x, s_0): f, ||c_e||_1 + ||c_i - s||_1, sum ln s, and the directional derivative of its
  // barrier cost from the point restoration started at (interior_point.hpp:729-752)
  double f_outer, viol_outer, logsum_outer, dphi_outer;
  // the smallest pivot of the rows eliminated in closed form at this iterate WITHOUT regularization (order p_e, n_e,
  // p_i, n_i): it joins the |D| >= 1e-4 test of the next iteration's unregularized attempt
  double eliminated_min_pivot;
};
struct FrHost {
  FrDirOut dir;
  IpmTrialOut trial;
  FrErrOut err;
  FrErrOut err_ahead;  // the same at the look-ahead iterate (expand(.., ahead))
};

class FrDevice {
 public:
  explicit FrDevice(DeviceNlp& dev);
  ~FrDevice();
  FrDevice(const FrDevice&) = delete;
  FrDevice& operator=(const FrDevice&) = delete;

  // Start of a restoration phase (host vectors): x_R and zeta D_R (n each), the outer problem's dense gradient and
  // slacks at the point of entry with its barrier parameter (for dphi_outer), the initial p, n with their slacks and
  // duals ([p_e | n_e | p_i | n_i], 2 m_e + 2 m_i each), and [1 | d_ce | d_ci] the error measure un-scales with.
  // x, s_0, y, z_0 are the outer system's buffers (upload_x / upload_duals).
  void begin(const double* x_r, const double* w, const double* g_outer, const double* s_outer, double mu_outer,
             const double* pn, const double* sx, const double* zx, const std::vector<double>& err_scales);

  // the reduced system for (delta; gamma is added by the factorization itself) -> the outer system's lhs / rhs.
  // soc: the second-order-correction right-hand side (interior_point.hpp:611-616); rhs_only: lhs is in place
  // second: into this object's own arrays (second_lhs() / second_rhs()) — the system of the attempt the regularization
  // policy would make next, factored beside the first in one launch (NewtonSystem::compute_hooked)
  void build(double delta, double mu, bool soc, bool rhs_only, bool second = false);
  // both of a launch's systems in ONE launch: (delta) into the outer system's arrays, (delta_second) into this object's
  void build_pair(double delta, double delta_second, double mu);
  const double* second_lhs() const { return m_lhs2.p; }
  const double* second_rhs() const { return m_rhs2.p; }
  // p = (dx, w) of the outer system -> the whole direction, its step sizes and directional derivative (-> host().dir,
  // alpha on the device), the first trial x (the outer system's trial input)
  // ahead: also the WHOLE iterate the full step (alpha_max, alpha_z) would give — x, s, y, z of all five blocks with the
  // z reset of interior_point.hpp:797-801 — into the look-ahead buffers (the outer system's and this object's): the
  // full tape and the error norms run on those next, and when the filter takes the point accept_lookahead() makes
  // them the current ones (ipm_kernels.h: ipm_lookahead_kernel, the same arrangement for the outer iteration)
  void expand(double delta, double mu, double tau, bool soc, bool ahead = false);
  void accept_lookahead();
  void trial_point(double alpha);                // trial x = x + alpha dx
  void trial_metrics(double alpha, double mu);   // after a value sweep at the trial x; alpha < 0: the device's alpha_max
  void commit(double alpha, double alpha_z, double mu);
  // after a full sweep at x -> host().err (err_ahead).  sums_ride: the sweep was made without its separable sums
  // (sweep_full(false)); they ride in this launch as extra workgroups, as in DeviceNlp::ipm_errors
  void errors(bool check_all_V, double mu, bool ahead = false, bool sums_ride = false);
  void soc_accumulate(double alpha, bool first);
  void save_direction();
  void restore_direction();
  void wait_published();
  const FrHost& host() const { return *m_host; }
  // [p_e | n_e | p_i | n_i], their slacks, their duals -> host
  void download_state(double* pn, double* sx, double* zx);
  void download_direction(double* dpn, double* psx, double* pzx);

  struct Args;  // the kernels' view (restoration.hip)

 private:
  Args args(bool ahead = false) const;
  DeviceNlp& m_dev;
  int m_n = 0, m_me = 0, m_mi = 0, m_M = 0;
  DevBuf<int32_t> m_diag_of;  // per lhs entry: the row whose diagonal it is, or -1
  DevBuf<double> m_xr, m_w, m_g_outer, m_s_outer, m_scales;
  double m_mu_outer = 0.0;
  DevBuf<double> m_pn, m_sx, m_zx, m_dpn, m_psx, m_pzx;
  DevBuf<double> m_pn_t, m_sx_t, m_zx_t;  // the look-ahead iterate's
  DevBuf<double> m_lhs2, m_rhs2;          // the second attempt's system (build(.., second))
  DevBuf<double> m_soc_ce, m_soc_c0, m_soc_x;
  DevBuf<double> m_keep_p, m_keep_ps0, m_keep_pz0, m_keep_dpn, m_keep_psx, m_keep_pzx;
  DevBuf<double> m_alpha, m_partial;
  DevBuf<unsigned int> m_done;
  DevBuf<double> m_expand_partial;       // expand(): the workgroups' shares of the four reductions
  DevBuf<unsigned int> m_expand_sync;    // ... workgroups through (word 0), the generation whose fold is done (word 16)
  unsigned int m_expand_generation = 0;
  DevBuf<unsigned long long> m_seq_dev;
  volatile unsigned long long* m_h_seq = nullptr;
  unsigned long long m_seq_expected = 0;
  FrHost* m_host = nullptr;
};

}  // namespace slpx
