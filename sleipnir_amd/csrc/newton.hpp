// NewtonSystem: one compiled NLP on one GPU — structure + plans + device state —
// and the host-side regularization policy that drives the device factorization.
//
// Counterpart of what interior_point() holds across iterations in the reference:
// the matrix callbacks (interior_point.hpp:199-237), `RegularizedLDLT solver`
// (:338-352) and the per-iteration Newton step (:426-482).  The δ/γ inertia-
// correction loop (util/sparse_regularized_ldlt.hpp:82-151) stays on the host so
// its decisions can be compared one-to-one with the oracle; every numeric
// factorization attempt inside it is a device launch.
#pragma once

#include <functional>
#include <future>
#include <limits>
#include <memory>
#include <vector>

#include "device.hpp"
#include "kkt_plan.hpp"
#include "ldlt_symbolic.hpp"
#include "nlp.hpp"

namespace slpx {

struct NewtonOptions {
  TapeCompileOptions tape;
  LdltOptions ldlt;
  int batch = 1;
  int device = 0;
  bool skip_structurally_singular_attempt = true;
};

// Eigen::ComputationInfo stand-in
enum class FactorInfo : int { Success = 0, NumericalIssue = 1 };

class FrDevice;

class NewtonSystem {
 public:
  NewtonSystem(Graph& g, const std::vector<NodeId>& x, NodeId f, const std::vector<NodeId>& c_e,
               const std::vector<NodeId>& c_i, const NewtonOptions& opt,
               const std::vector<int32_t>* user_perm = nullptr, bool defer_device = false);
  // `defer_device`: the constructor stops after the host part — AD structure, tapes, KKT plan, symbolic
  // factorization: everything that reads the expression graph — and finish_device() does the rest (device
  // memory, uploads, the generated tape kernel: from the caches or hipRTC) when the system is first needed.
  bool has_device() const { return m_dev != nullptr; }
  void finish_device();

  // Linear-solver seam only (RegularizedLDLT, util/regularized_ldlt.hpp:45-51, sparse
  // branch): no expression graph, no tape, no KKT assembly — just the lower-triangular CSC
  // pattern of the matrix compute() will be given (diagonal entries may be absent; they are
  // added, sparse_regularized_ldlt.hpp:67).  `n_dec` leading rows/columns get +delta, the
  // remaining `m_e` get -gamma.
  NewtonSystem(const CscPattern& lower, int n_dec, int m_e, const NewtonOptions& opt);
  // position in the internal lhs (pattern 5) of every entry of the pattern given above
  const std::vector<int32_t>& user_lhs_map() const { return m_user_lhs_map; }

  // The model this system was compiled from (feasibility restoration builds its
  // augmented model out of the same expressions).
  Graph& graph() const { return *m_graph; }
  const std::vector<NodeId>& x_nodes() const { return m_x_nodes; }
  const std::vector<NodeId>& c_e_nodes() const { return m_ce_nodes; }
  const std::vector<NodeId>& c_i_nodes() const { return m_ci_nodes; }
  const NewtonOptions& options() const { return m_opt; }

  // Feasibility restoration runs on THIS system (restoration.hpp): the device state of the restoration iterate
  // beyond (x, s, y, z), made when a solve first enters restoration
  FrDevice& restoration_device();
  ~NewtonSystem();

  const NlpStructure& structure() const { return m_s; }
  const KktPlan& kkt() const { return m_k; }
  const LdltPlan& ldlt() const { return m_l; }
  DeviceNlp& device() { return *m_dev; }
  int batch() const { return m_opt.batch; }

  void set_gamma_min(double g) { m_gamma_min = g; }
  void reset_regularization();
  using RegularizationState = std::pair<std::vector<double>, std::vector<double>>;
  RegularizationState regularization_state() const { return {m_prev_delta, m_prev_gamma}; }
  void set_regularization_state(const RegularizationState& st) {
    m_prev_delta = st.first;
    m_prev_gamma = st.second;
  }

  // sparse_regularized_ldlt.hpp:64-152 on the lhs currently in device memory.
  // Returns per-problem info; fills the regularization that was used.
  // solve_speculatively: also run solve() + backsub() after every attempt (see newton.cpp).
  std::vector<FactorInfo> compute(bool solve_speculatively = false);
  std::vector<FactorInfo> compute_impl(int mode);
  // One factorization of the lhs in device memory with delta = gamma = 0, accepted whenever it
  // has no zero / non-finite pivot and the ideal inertia — no |D| threshold, no regularization
  // memory touched.  For systems that are not KKT systems of the barrier problem: the
  // least-squares multiplier estimate, which the reference factors unregularized
  // (lagrange_multiplier_estimate.hpp:107).  Single problem.
  bool factor_unregularized();
  // The same policy loop (sparse_regularized_ldlt.hpp:64-152) for a caller that writes the system of every attempt
  // itself — feasibility restoration, whose extra variables are eliminated in closed form (restoration.hpp): their part
  // of the system depends on delta.  `prepare(delta, gamma)` leaves lhs / rhs of the attempt in device memory,
  // `after(delta, gamma)` is enqueued behind its solve, `eliminated_min_pivot()` (first attempt only) is the smallest
  // pivot of what was eliminated outside the factorization: it joins the |D| >= 1e-4 test (:82-87).  The
  // unregularized attempt is always made (such a system has no structurally zero pivot).  One problem.
  // `prepare_second(delta, gamma, &lhs2, &rhs2)` (optional): the system of the attempt the policy would make NEXT, in
  // buffers of the caller's — the two are then factored in ONE launch where the device can (ldlt_mf_twin_kernel, as
  // compute() does for the systems it evaluates itself), judged in the policy's order from their own counters: the
  // sequence of (delta, gamma) tried, the one accepted and the count of factorizations are the sequential loop's.
  // `after` then follows only the FIRST attempt of a launch (a direction taken from the second is the caller's to
  // expand after compute_hooked returns: last_hooked_chain_valid()).
  struct AttemptHooks {
    std::function<void(double, double)> prepare, after;
    std::function<double()> eliminated_min_pivot;
    std::function<void(double, double, const double**, const double**)> prepare_second;
    // (optional, beside prepare_second) both systems of a launch at once: (d0, g0, d1, g1, &lhs2, &rhs2)
    std::function<void(double, double, double, double, const double**, const double**)> prepare_pair;
  };
  bool last_hooked_chain_valid() const { return m_hooked_chain_valid; }
  std::vector<FactorInfo> compute_hooked(const AttemptHooks& hooks);
  const std::vector<double>& hessian_regularization() const { return m_prev_delta; }
  const std::vector<double>& constraint_jacobian_regularization() const { return m_prev_gamma; }
  int last_factorizations() const { return m_last_factorizations; }
  // Work enqueued behind every speculative solve of compute(true) — BEFORE the host reads
  // the inertia counters, so it costs no extra synchronization when the attempt is accepted
  // (the interior-point driver puts its step-size / trial-point kernels here).
  void set_after_attempt(std::function<void()> fn) { m_after_attempt = std::move(fn); }
  // Twin attempts (DeviceNlp::factor_solve_publish_twin): compute(solve_speculatively) factors the policy's
  // attempt and the one that would follow it in one launch.  Only for a caller whose after_attempt work takes the
  // direction of whichever attempt the policy takes ON THE DEVICE (DeviceNlp::ipm_lookahead does), and whose system is
  // the one the device's V, s, y, z describe (build_kkt_for_step — later launches of the loop evaluate it again).
  void set_twin_attempts(bool on) { m_twin_attempts = on; }
  // The first launch of the NEXT compute(true) — the policy's attempt and its twin, with the after_attempt chain behind
  // it — enqueued now, BEFORE the host knows whether that step will be wanted: the launch waits for the word the error
  // launch in front of it leaves (DeviceNlp::ipm_gate_next_step) and passes if the iteration was not decided on the
  // device.  The compute(true) that follows takes the launch as its first; cancel_speculative_compute() puts the
  // launch bookkeeping back when the device let it pass.  false: not possible now (no launch was made).
  // gated: the launch reads the word the deciding error launch in front of it leaves and passes if told to; false: it
  // carries that error launch itself (DeviceNlp::ipm_ride_errors_in_next_step was armed) and holds its results back
  // instead — cancel_speculative_compute(launch_ran = true) then.
  bool begin_speculative_compute(bool gated = true);
  void cancel_speculative_compute(bool launch_ran = false);
  bool speculative_compute_pending() const { return m_spec.valid; }
  // launches with two attempts since construction, by what the first attempt showed: accepted; the failure the
  // second attempt stood for, and the second accepted / not; zero pivots; the other inertia failure; the
  // factorization itself failed
  const long* twin_histogram() const { return m_twin_hist; }
  int last_twin_launches() const { return m_last_twin_launches; }   // step launches of the last compute() that held two attempts
  int last_twin_taken() const { return m_last_twin_taken; }         // ... whose second attempt the policy took

  // One full Newton step on device-resident state: AD refresh, KKT lhs/rhs,
  // regularized factorization, solve, back-substitution
  // (interior_point.hpp:809-812 + :426-482).
  std::vector<FactorInfo> newton_step(bool refresh_ad = true);

 private:
  NewtonOptions m_opt;
  Graph* m_graph = nullptr;
  std::vector<NodeId> m_x_nodes, m_ce_nodes, m_ci_nodes;
  std::unique_ptr<FrDevice> m_fr;
  NlpStructure m_s;
  KktPlan m_k;
  LdltPlan m_l;
  std::unique_ptr<DeviceNlp> m_dev;
  double m_gamma_min = 1e-10;
  std::vector<double> m_prev_delta, m_prev_gamma;
  int m_last_factorizations = 0;
  std::function<void()> m_after_attempt;
  bool m_twin_attempts = false;
  int m_last_twin_launches = 0, m_last_twin_taken = 0;
  bool m_hooked_chain_valid = false;  // compute_hooked: `after` ran behind the attempt that was accepted
  long m_twin_hist[6] = {0, 0, 0, 0, 0, 0};
  int m_twin_expect = 1;  // what the loop's first attempt drew last time (compute_twin): 1 negative pivots, 3 positive
  std::vector<FactorInfo> compute_twin();
  struct TwinLaunch {
    double d0, g0, d1, g1;
    int mode;
  };
  TwinLaunch twin_first_launch() const;  // what compute_twin's first launch is, from the policy's memory
  struct Speculative {
    bool valid = false, have_second = false;
    TwinLaunch launch{};
    DeviceNlp::LaunchBook book{};
  } m_spec;
  std::vector<int32_t> m_user_lhs_map;
};

}  // namespace slpx
