// Host driver of the interior-point iteration around the device Newton step.
//
// Mirrors include/sleipnir/optimization/solver/interior_point.hpp:63-878 step for
// step (same constants, same filter line search, same second-order corrections,
// same barrier update and exits); every O(nnz) piece of the Newton step itself
// (:426-482, :809-812) runs on the GPU through NewtonSystem, and so does every O(n)
// piece of the iteration around it (step sizes, trial iterate, filter quantities,
// second-order corrections, iterate update, error norms: ipm_kernels.h, SURVEY.md §8f rows
// N1/N2).  The DECISIONS (filter, barrier update, exits) are functions of a few dozen scalars
// per iteration (ipm_decide.h): the common iteration's — the filter takes the full step, the
// error is above the tolerance, the barrier parameter stays — are taken on the device by the
// launch that reduces the norms, with the next step enqueued behind it (SLPX_IPM_PIPELINE=0:
// by the host); every other iteration's by the host.  SLPX_IPM_RESIDENT=0 selects the older
// driver that keeps the vectors on the host (the cross-check of the resident one).
//
// Feasibility restoration (util/feasibility_restoration.hpp:347-628, row N3) runs on the SAME
// compiled NewtonSystem: its extra variables p, n are eliminated from the Newton-KKT system in
// closed form (restoration.hpp), what is left has the outer problem's pattern;
// the least-squares multiplier estimate (util/lagrange_multiplier_estimate.hpp:56-133)
// that ends it is one more factorization on the outer system's KKT pattern.
#pragma once


#include <functional>
#include <limits>
#include <vector>

#include "newton.hpp"

namespace slpx {

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

// iteration_info.hpp:13-41 (matrices as value arrays over the static patterns)
struct IterationInfo {
  int iteration;
  const std::vector<double>& x;
  const std::vector<double>& s;
  const std::vector<double>& y;
  const std::vector<double>& z;
  const std::vector<double>& V;  // [f | c_e | c_i | g | A_e | A_i | H_f | H_c], see nlp.hpp
  // The system these arrays belong to: inside feasibility restoration that is the restoration
  // model (n + 2 m_e + 2 m_i variables, m_i + 2 m_e + 2 m_i inequality rows; the outer
  // problem's x, s come first), not the user's problem.
  const NlpStructure* structure = nullptr;
  bool in_restoration = false;
};
using IterationCallback = std::function<bool(const IterationInfo&)>;

struct SolveReport {
  int iterations = 0;
  int factorizations = 0;
  int solves = 0;
  int value_sweeps = 0;
  int restorations = 0;
  int restoration_iterations = 0;  // iterations spent inside feasibility restoration
  double delta = 0.0, gamma = 0.0;
  double final_error = 0.0;
  // wall-clock per phase, seconds (names follow interior_point.hpp:155-174)
  double t_setup = 0, t_kkt_build = 0, t_kkt_decomp = 0, t_kkt_solve = 0, t_line_search = 0,
         t_ad_refresh = 0, t_total = 0;
  // feasibility restoration: allocating its device state (first phase only), and the
  // restoration iterations themselves (both are part of t_total)
  double t_restoration_setup = 0, t_restoration = 0;
};

// Scaling at x0 (util/problem_scaling.hpp:100-107) from an unscaled V.
std::vector<double> compute_problem_scaling(const NlpStructure& s, const std::vector<double>& V_raw);

// The reference's solvers for problems without inequality constraints, on the same device
// Newton step: sqp() (solver/sqp.hpp:98-604; equality constraints only, problem.hpp:403) and
// newton() (solver/newton.hpp:51-292; unconstrained, problem.hpp:335).  x in/out.
ExitStatus sqp(NewtonSystem& sys, const std::vector<double>& scales, const std::vector<IterationCallback>& callbacks,
               const Options& options, std::vector<double>& x, std::vector<double>* y_out = nullptr,
               SolveReport* report = nullptr);
ExitStatus newton(NewtonSystem& sys, const std::vector<double>& scales,
                  const std::vector<IterationCallback>& callbacks, const Options& options,
                  std::vector<double>& x, SolveReport* report = nullptr);

// feasibility_restoration (util/feasibility_restoration.hpp:347-628) on its own, from a given
// iterate: sets the restoration problem up around (x, s), runs `steps` iterations of its
// interior-point loop (a callback then stops it, the reference's own way out, :729-752) and, as
// after any accepted restoration, replaces y, z by the least-squares multiplier estimate
// (lagrange_multiplier_estimate.hpp:56-133).  `scales` must be installed on the device.
ExitStatus feasibility_restoration_steps(NewtonSystem& sys, const std::vector<double>& scales,
                                         const Options& options, std::vector<double>& x,
                                         std::vector<double>& s, std::vector<double>& y,
                                         std::vector<double>& z, double mu, int steps,
                                         SolveReport* report = nullptr);

// x in/out.  `scales` = [d_f, d_ce.., d_ci..] already installed on the device.
ExitStatus interior_point(NewtonSystem& sys, const std::vector<double>& scales,
                          const std::vector<IterationCallback>& callbacks, const Options& options,
                          std::vector<double>& x, std::vector<double>* s_out = nullptr,
                          std::vector<double>* y_out = nullptr, std::vector<double>* z_out = nullptr,
                          SolveReport* report = nullptr);

}  // namespace slpx
