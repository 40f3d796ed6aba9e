// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The minimum-time differential-drive problem of the reference's OCP test
// (test/src/optimization/differential_drive_ocp_test.cpp:24-107 with the model of
// test/include/differential_drive_util.hpp): 50 steps, ONE shared timestep variable
// (TimestepMethod::VARIABLE_SINGLE), direct transcription of an explicit ODE, drive from the
// origin to (1, 1) at rest.  Same checks: LINEAR cost, NONLINEAR equalities, LINEAR
// inequalities, SUCCESS, initial and final state to 1e-8.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <exception>

#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/ocp.hpp>

namespace {
constexpr double trackwidth = 0.699, Kv_linear = 3.02, Ka_linear = 0.642, Kv_angular = 1.382, Ka_angular = 0.08495;
constexpr double A1 = -(Kv_linear / Ka_linear + Kv_angular / Ka_angular) / 2.0;
constexpr double A2 = -(Kv_linear / Ka_linear - Kv_angular / Ka_angular) / 2.0;
constexpr double B1 = 0.5 / Ka_linear + 0.5 / Ka_angular;
constexpr double B2 = 0.5 / Ka_linear - 0.5 / Ka_angular;

// x = [x, y, heading, left velocity, right velocity], u = [left voltage, right voltage]
slp::VariableMatrix<double> dynamics(const slp::VariableMatrix<double>& x, const slp::VariableMatrix<double>& u) {
  slp::VariableMatrix<double> xdot{5};
  auto v = (x[3] + x[4]) / 2.0;
  xdot[0] = v * cos(x[2]);
  xdot[1] = v * sin(x[2]);
  xdot[2] = (x[4] - x[3]) / trackwidth;
  slp::DenseMatrix A{{A1, A2}, {A2, A1}};
  slp::DenseMatrix B{{B1, B2}, {B2, B1}};
  xdot.segment(3, 2) = A * x.segment(3, 2) + B * u;
  return xdot;
}
}  // namespace

int main(int argc, char**) {
  constexpr int N = 50;
  const std::chrono::duration<double> min_timestep{0.05};
  try {
    slp::OCP<double> problem(5, 2, min_timestep, N, dynamics, slp::DynamicsType::EXPLICIT_ODE,
                             slp::TimestepMethod::VARIABLE_SINGLE, slp::TranscriptionMethod::DIRECT_TRANSCRIPTION);
    // seed the minimum-time formulation with a straight line between the waypoints
    for (int i = 0; i < N + 1; ++i) {
      problem.X()[0, i].set_value(static_cast<double>(i) / (N + 1));
      problem.X()[1, i].set_value(static_cast<double>(i) / (N + 1));
    }
    slp::DenseMatrix x_initial{{0.0}, {0.0}, {0.0}, {0.0}, {0.0}};
    slp::DenseMatrix x_final{{1.0}, {1.0}, {0.0}, {0.0}, {0.0}};
    slp::DenseMatrix u_min{{-12.0}, {-12.0}};
    slp::DenseMatrix u_max{{12.0}, {12.0}};
    problem.constrain_initial_state(x_initial);
    problem.constrain_final_state(x_final);
    problem.set_lower_input_bound(u_min);
    problem.set_upper_input_bound(u_max);
    problem.set_min_timestep(min_timestep);
    problem.set_max_timestep(std::chrono::duration<double>{3.0});

    slp::DenseMatrix ones{N + 1, 1};
    for (int k = 0; k < N + 1; ++k) ones[k, 0] = 1.0;
    problem.minimize(problem.dt() * ones);

    std::printf("cost=%d eq=%d ineq=%d\n", static_cast<int>(problem.cost_function_type()),
                static_cast<int>(problem.equality_constraint_type()),
                static_cast<int>(problem.inequality_constraint_type()));
    if (argc > 1) return 0;  // model only (no device needed)

    const auto status = problem.solve();
    int bad = static_cast<int>(status) != 0;
    auto X = problem.X();
    for (int r = 0; r < 5; ++r) {
      bad += !(std::abs(X.value(r, 0) - x_initial[r, 0]) < 1e-8);
      bad += !(std::abs(X.value(r, N) - x_final[r, 0]) < 1e-8);
    }
    std::printf("status=%d total time %.6f s (dt %.6f) final=(%.9f, %.9f, %.2e) failed_checks=%d\n",
                static_cast<int>(status), problem.dt().value(0, 0) * N, problem.dt().value(0, 0), X.value(0, N),
                X.value(1, N), X.value(2, N), bad);
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
