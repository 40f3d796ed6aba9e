// TEST / BENCH FIXTURE — not part of the product (libslpx.so holds no model).
//
// The two models of the reference's scalability benchmarks as USER programs of the slp:: surface
// (reference: benchmarks/scalability/cart_pole/sleipnir.cpp:16-129, benchmarks/scalability/flywheel/
// sleipnir.cpp:12-42, benchmarks/rk4.hpp:14-23).  The operations are recorded in the order the
// benchmark programs record them, so that the expression graph — and with it every rounding of a
// solve — is the graph the reference's own program builds; Eigen constants are slp::DenseMatrix.
// bench_models.cpp hands the finished model to the product through the public C-ABI
// (slpx_problem_create / slpx_problem_adopt_variable / slpx_problem_subject_to_*).
#pragma once

#include <chrono>
#include <cmath>
#include <numbers>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/problem.hpp>

namespace bench_models {

using Mat = slp::VariableMatrix<double>;

// one classical Runge-Kutta step (template parameter list of benchmarks/rk4.hpp)
template <typename F, typename T, typename U>
T rk4(F&& f, T x, U u, std::chrono::duration<double> dt) {
  const auto h = dt.count();
  T k1 = f(x, u);
  T k2 = f(x + h * 0.5 * k1, u);
  T k3 = f(x + h * 0.5 * k2, u);
  T k4 = f(x + h * k3, u);
  return x + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
}

// M(q) q'' + C(q, q') q' = tau_g(q) + B u;  x = [position, angle, their rates]
inline Mat cart_pole_dynamics(const Mat& x, const Mat& u) {
  constexpr double m_c = 5.0, m_p = 0.5, l = 0.5, g = 9.806;  // kg, kg, m, m/s²
  auto q = x.segment(0, 2);
  auto qdot = x.segment(2, 2);
  auto theta = q[1];
  auto thetadot = qdot[1];

  Mat M{{m_c + m_p, m_p * l * cos(theta)}, {m_p * l * cos(theta), m_p * std::pow(l, 2)}};
  Mat C{{0, -m_p * l * thetadot * sin(theta)}, {0, 0}};
  Mat tau_g{{0}, {-m_p * g * l * sin(theta)}};
  slp::DenseMatrix B{{1}, {0}};

  Mat qddot{4, 1};
  qddot.segment(0, 2) = qdot;
  qddot.segment(2, 2) = solve(M, tau_g - C * qdot + B * u);
  return qddot;
}

// swing-up over N steps of dt seconds: bounds on the cart position and the force, RK4 defects
inline void build_cart_pole(slp::Problem<double>& problem, double dt_seconds, int N) {
  const std::chrono::duration<double> dt{dt_seconds};
  constexpr double u_max = 20.0, d_max = 2.0;  // N, m
  const slp::DenseMatrix x_initial = slp::DenseMatrix::vector({0.0, 0.0, 0.0, 0.0});
  const slp::DenseMatrix x_final = slp::DenseMatrix::vector({1.0, std::numbers::pi, 0.0, 0.0});

  auto X = problem.decision_variable(4, N + 1);
  for (int k = 0; k < N + 1; ++k) {
    X[0, k].set_value(std::lerp(x_initial[0], x_final[0], static_cast<double>(k) / N));
    X[1, k].set_value(std::lerp(x_initial[1], x_final[1], static_cast<double>(k) / N));
  }
  auto U = problem.decision_variable(1, N);

  problem.subject_to(X.col(0) == x_initial);
  problem.subject_to(X.col(N) == x_final);
  problem.subject_to(slp::bounds(0.0, X.row(0), d_max));
  problem.subject_to(slp::bounds(-u_max, U, u_max));
  for (int k = 0; k < N; ++k)
    problem.subject_to(X.col(k + 1) == rk4<decltype(cart_pole_dynamics), Mat, Mat>(cart_pole_dynamics, X.col(k), U.col(k), dt));

  slp::Variable J = 0.0;
  for (int k = 0; k < N; ++k) J += U.col(k).T() * U.col(k);
  problem.minimize(J);
}

// discrete first-order lag tracking r = 10 with |u| <= 12
inline void build_flywheel(slp::Problem<double>& problem, double dt, int N) {
  slp::DenseMatrix A{{std::exp(-dt)}};
  slp::DenseMatrix B{{1.0 - std::exp(-dt)}};

  auto X = problem.decision_variable(1, N + 1);
  auto U = problem.decision_variable(1, N);
  for (int k = 0; k < N; ++k) problem.subject_to(X.col(k + 1) == A * X.col(k) + B * U.col(k));
  problem.subject_to(X.col(0) == 0.0);
  problem.subject_to(slp::bounds(-12, U, 12));

  slp::DenseMatrix r{{10.0}};
  slp::Variable J = 0.0;
  for (int k = 0; k < N + 1; ++k) J += ((r - X.col(k)).T() * (r - X.col(k)));
  problem.minimize(J);
}

}  // namespace bench_models
