// The reference's benchmark models written against the slp:: surface, statement for
// statement as in the reference sources (so the recorded graphs are identical):
//   benchmarks/scalability/cart_pole/sleipnir.cpp:16-129
//   benchmarks/scalability/flywheel/sleipnir.cpp:12-42
//   benchmarks/rk4.hpp:14-23
// Same spellings as there (slp::Problem<double>, slp::VariableMatrix<double>, rk4<...>); the only
// substitution is Eigen::Matrix / Eigen::Vector constants -> slp::DenseMatrix (Eigen is not in
// this toolchain).  tests/test_slp_surface.py compiles the same text as a user program.
#include "problems.hpp"

#include <chrono>
#include <cmath>
#include <numbers>

namespace slpx_models {

using slp::DenseMatrix;

// benchmarks/rk4.hpp:14-23
template <typename F, typename T, typename U>
T rk4(F&& f, T x, U u, std::chrono::duration<double> dt) {
  const auto h = dt.count();

  T k1 = f(x, u);
  T k2 = f(x + h * 0.5 * k1, u);
  T k3 = f(x + h * 0.5 * k2, u);
  T k4 = f(x + h * k3, u);

  return x + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
}

// cart_pole/sleipnir.cpp:16-74
slp::VariableMatrix<double> cart_pole_dynamics(const slp::VariableMatrix<double>& x,
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
  DenseMatrix B{{1}, {0}};

  slp::VariableMatrix<double> qddot{4, 1};
  qddot.segment(0, 2) = qdot;
  qddot.segment(2, 2) = solve(M, tau_g - C * qdot + B * u);
  return qddot;
}

// cart_pole/sleipnir.cpp:76-129
void build_cart_pole(slp::Problem<double>& problem, double dt_seconds, int N, slp::VariableMatrix<double>* X_out,
                     slp::VariableMatrix<double>* U_out) {
  const std::chrono::duration<double> dt{dt_seconds};
  constexpr double u_max = 20.0;  // N
  constexpr double d_max = 2.0;   // m
  const DenseMatrix x_initial = DenseMatrix::vector({0.0, 0.0, 0.0, 0.0});
  const DenseMatrix x_final = DenseMatrix::vector({1.0, std::numbers::pi, 0.0, 0.0});

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

  for (int k = 0; k < N; ++k) {
    problem.subject_to(X.col(k + 1) ==
                       rk4<decltype(cart_pole_dynamics), slp::VariableMatrix<double>, slp::VariableMatrix<double>>(
                           cart_pole_dynamics, X.col(k), U.col(k), dt));
  }

  slp::Variable J = 0.0;
  for (int k = 0; k < N; ++k) {
    J += U.col(k).T() * U.col(k);
  }
  problem.minimize(J);
  if (X_out) *X_out = X;
  if (U_out) *U_out = U;
}

// flywheel/sleipnir.cpp:12-42
void build_flywheel(slp::Problem<double>& problem, double dt, int N, slp::VariableMatrix<double>* X_out,
                    slp::VariableMatrix<double>* U_out) {
  DenseMatrix A{{std::exp(-dt)}};
  DenseMatrix B{{1.0 - std::exp(-dt)}};

  auto X = problem.decision_variable(1, N + 1);
  auto U = problem.decision_variable(1, N);

  for (int k = 0; k < N; ++k) {
    problem.subject_to(X.col(k + 1) == A * X.col(k) + B * U.col(k));
  }
  problem.subject_to(X.col(0) == 0.0);
  problem.subject_to(slp::bounds(-12, U, 12));

  DenseMatrix r{{10.0}};
  slp::Variable J = 0.0;
  for (int k = 0; k < N + 1; ++k) {
    J += ((r - X.col(k)).T() * (r - X.col(k)));
  }
  problem.minimize(J);
  if (X_out) *X_out = X;
  if (U_out) *U_out = U;
}

}  // namespace slpx_models
