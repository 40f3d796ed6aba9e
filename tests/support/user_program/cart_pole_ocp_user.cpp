// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The cart-pole swing-up of the reference's OCP test (test/src/optimization/cart_pole_ocp_test.cpp:
// 28-96, model of test/include/cart_pole_util.hpp = the benchmark's): 100 steps, ONE shared
// timestep variable, Hermite-Simpson direct collocation, cart position bounds through
// for_each_step.  Same checks: QUADRATIC cost, NONLINEAR equalities, LINEAR inequalities,
// SUCCESS, initial and final state to 1e-8.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <exception>
#include <numbers>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/ocp.hpp>

slp::VariableMatrix<double> cart_pole_dynamics(
    const slp::VariableMatrix<double>& x,
    const slp::VariableMatrix<double>& u) {
  constexpr double m_c = 5.0;  // Cart mass (kg)
  constexpr double m_p = 0.5;  // Pole mass (kg)
  constexpr double l = 0.5;    // Pole length (m)
  constexpr double g = 9.806;  // Acceleration due to gravity (m/s²)

  auto q = x.segment(0, 2);
  auto qdot = x.segment(2, 2);
  auto theta = q[1];
  auto thetadot = qdot[1];

  slp::VariableMatrix<double> M{{m_c + m_p, m_p * l * cos(theta)},
                                {m_p * l * cos(theta), m_p * std::pow(l, 2)}};
  slp::VariableMatrix<double> C{{0, -m_p * l * thetadot * sin(theta)}, {0, 0}};
  slp::VariableMatrix<double> tau_g{{0}, {-m_p * g * l * sin(theta)}};
  slp::DenseMatrix B{{1}, {0}};  // Eigen::Matrix<double, 2, 1> B{{1}, {0}};

  slp::VariableMatrix<double> qddot{4, 1};
  qddot.segment(0, 2) = qdot;
  qddot.segment(2, 2) = solve(M, tau_g - C * qdot + B * u);
  return qddot;
}

int main(int argc, char**) {
  constexpr std::chrono::duration<double> dt{0.05};
  constexpr int N = 100;
  constexpr double u_max = 20.0;  // N
  constexpr double d_max = 2.0;   // m
  const slp::DenseMatrix x_initial = slp::DenseMatrix::vector({0.0, 0.0, 0.0, 0.0});
  const slp::DenseMatrix x_final = slp::DenseMatrix::vector({1.0, std::numbers::pi, 0.0, 0.0});
  try {
    slp::OCP<double> problem(4, 1, dt, N, cart_pole_dynamics, slp::DynamicsType::EXPLICIT_ODE,
                             slp::TimestepMethod::VARIABLE_SINGLE, slp::TranscriptionMethod::DIRECT_COLLOCATION);
    auto& X = problem.X();
    for (int k = 0; k < N + 1; ++k) {
      X[0, k].set_value(std::lerp(x_initial[0], x_final[0], static_cast<double>(k) / N));
      X[1, k].set_value(std::lerp(x_initial[1], x_final[1], static_cast<double>(k) / N));
    }
    problem.constrain_initial_state(x_initial);
    problem.constrain_final_state(x_final);
    problem.for_each_step([&](const slp::VariableMatrix<double>& x, const slp::VariableMatrix<double>&) {
      problem.subject_to(slp::bounds(0.0, x[0], d_max));
    });
    problem.set_lower_input_bound(-u_max);
    problem.set_upper_input_bound(u_max);
    auto& U = problem.U();
    slp::Variable J = 0.0;
    for (int k = 0; k < N; ++k) J += U.col(k).T() * U.col(k);
    problem.minimize(J);

    std::printf("cost=%d eq=%d ineq=%d\n", static_cast<int>(problem.cost_function_type()),
                static_cast<int>(problem.equality_constraint_type()),
                static_cast<int>(problem.inequality_constraint_type()));
    if (argc > 1) return 0;  // model only (no device needed)

    const auto status = problem.solve();
    int bad = static_cast<int>(status) != 0;
    for (int r = 0; r < 4; ++r) {
      bad += !(std::abs(problem.X().value(r, 0) - x_initial[r]) < 1e-8);
      bad += !(std::abs(problem.X().value(r, N) - x_final[r]) < 1e-8);
    }
    std::printf("status=%d iterations=%d dt=%.6f final=(%.9f, %.9f) failed_checks=%d\n", static_cast<int>(status),
                problem.report().iterations, problem.dt().value(0, 0), problem.X().value(0, N), problem.X().value(1, N),
                bad);
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
