// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// The reference's benchmark problem definitions, restated on the oracle DSL:
//   benchmarks/scalability/cart_pole/sleipnir.cpp:16-129
//   benchmarks/scalability/flywheel/sleipnir.cpp:12-42
//   benchmarks/rk4.hpp:14-23
#pragma once

#include <cmath>

#include "problem.hpp"

namespace orc {

// benchmarks/rk4.hpp:14-23
template <typename F>
VarMat rk4(F&& f, const VarMat& x, const VarMat& u, double h) {
  VarMat k1 = f(x, u);
  VarMat k2 = f(x + h * 0.5 * k1, u);
  VarMat k3 = f(x + h * 0.5 * k2, u);
  VarMat k4 = f(x + h * k3, u);
  return x + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
}

// cart_pole/sleipnir.cpp:16-74
inline VarMat cart_pole_dynamics(const VarMat& x, const VarMat& u) {
  constexpr double m_c = 5.0;
  constexpr double m_p = 0.5;
  constexpr double l = 0.5;
  constexpr double g = 9.806;

  VarMat q = x.segment(0, 2);
  VarMat qdot = x.segment(2, 2);
  Var theta = q(1);
  Var thetadot = qdot(1);

  VarMat M{{m_c + m_p, Var(m_p * l) * cos(theta)}, {Var(m_p * l) * cos(theta), m_p * std::pow(l, 2)}};
  VarMat C{{0.0, Var(-m_p * l) * thetadot * sin(theta)}, {0.0, 0.0}};
  VarMat tau_g{{0.0}, {Var(-m_p * g * l) * sin(theta)}};
  Mat B{{1.0}, {0.0}};

  VarMat qddot(4, 1);
  qddot.set_block(0, 0, qdot);
  qddot.set_block(2, 0, solve(M, tau_g - C * qdot + B * u));
  return qddot;
}

// cart_pole/sleipnir.cpp:76-129
struct CartPole {
  Problem problem;
  VarMat X, U;
};

inline void build_cart_pole(CartPole& cp, double dt, int N) {
  constexpr double u_max = 20.0;
  constexpr double d_max = 2.0;
  const double x_initial[4] = {0.0, 0.0, 0.0, 0.0};
  const double x_final[4] = {1.0, M_PI, 0.0, 0.0};
  auto lerp = [](double a, double b, double t) { return a + t * (b - a); };

  Problem& problem = cp.problem;
  cp.X = problem.decision_variable(4, N + 1);
  VarMat& X = cp.X;
  for (int k = 0; k < N + 1; ++k) {
    X(0, k).set_value(lerp(x_initial[0], x_final[0], static_cast<double>(k) / N));
    X(1, k).set_value(lerp(x_initial[1], x_final[1], static_cast<double>(k) / N));
  }
  cp.U = problem.decision_variable(1, N);
  VarMat& U = cp.U;

  problem.subject_to_eq(eq(X.col(0), col_vector({x_initial[0], x_initial[1], x_initial[2], x_initial[3]})));
  problem.subject_to_eq(eq(X.col(N), col_vector({x_final[0], x_final[1], x_final[2], x_final[3]})));
  problem.subject_to_ineq(bounds(VarMat(Var(0.0)), X.row(0), VarMat(Var(d_max))));
  problem.subject_to_ineq(bounds(VarMat(Var(-u_max)), U, VarMat(Var(u_max))));

  for (int k = 0; k < N; ++k) {
    problem.subject_to_eq(eq(X.col(k + 1), rk4(cart_pole_dynamics, X.col(k), U.col(k), dt)));
  }

  Var J = 0.0;
  for (int k = 0; k < N; ++k) {
    J += Var((U.col(k).T() * U.col(k))(0, 0));
  }
  problem.minimize(J);
}

// flywheel/sleipnir.cpp:12-42
struct Flywheel {
  Problem problem;
  VarMat X, U;
};

inline void build_flywheel(Flywheel& fw, double dt, int N) {
  Mat A{{std::exp(-dt)}};
  Mat B{{1.0 - std::exp(-dt)}};
  Problem& problem = fw.problem;
  fw.X = problem.decision_variable(1, N + 1);
  fw.U = problem.decision_variable(1, N);
  VarMat& X = fw.X;
  VarMat& U = fw.U;
  for (int k = 0; k < N; ++k) {
    problem.subject_to_eq(eq(X.col(k + 1), A * X.col(k) + B * U.col(k)));
  }
  problem.subject_to_eq(eq(X.col(0), VarMat(Var(0.0))));
  problem.subject_to_ineq(bounds(VarMat(Var(-12)), U, VarMat(Var(12))));
  VarMat r{{10.0}};
  Var J = 0.0;
  for (int k = 0; k < N + 1; ++k) {
    J += Var(((r - X.col(k)).T() * (r - X.col(k)))(0, 0));
  }
  problem.minimize(J);
}

}  // namespace orc
