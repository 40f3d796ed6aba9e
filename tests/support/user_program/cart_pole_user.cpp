// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The cart-pole model of the reference's scalability benchmark
// (benchmarks/scalability/cart_pole/sleipnir.cpp:16-129, benchmarks/rk4.hpp:14-23) with the
// reference's own include lines and spellings — slp::VariableMatrix<double>,
// slp::Problem<double>, problem.decision_variable(4, N + 1), X[0, k], slp::bounds,
// solve(M, ...), rk4<decltype(f), slp::VariableMatrix<double>, slp::VariableMatrix<double>> —
// compiled against <repo>/include and linked with libslpx.so by tests/test_slp_surface.py.
// The one substitution: Eigen::Matrix / Eigen::Vector constants are slp::DenseMatrix (Eigen is
// not in this toolchain).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <numbers>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/problem.hpp>

template <typename F, typename T, typename U>
T rk4(F&& f, T x, U u, std::chrono::duration<double> dt) {
  const auto h = dt.count();

  T k1 = f(x, u);
  T k2 = f(x + h * 0.5 * k1, u);
  T k3 = f(x + h * 0.5 * k2, u);
  T k4 = f(x + h * k3, u);

  return x + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
}

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

slp::Problem<double> cart_pole_sleipnir(std::chrono::duration<double> dt,
                                        int N) {
  constexpr double u_max = 20.0;  // N
  constexpr double d_max = 2.0;   // m

  const slp::DenseMatrix x_initial = slp::DenseMatrix::vector({0.0, 0.0, 0.0, 0.0});
  const slp::DenseMatrix x_final = slp::DenseMatrix::vector({1.0, std::numbers::pi, 0.0, 0.0});

  slp::Problem<double> problem;

  // x = [q, q̇]ᵀ = [x, θ, ẋ, θ̇]ᵀ
  auto X = problem.decision_variable(4, N + 1);

  // Initial guess
  for (int k = 0; k < N + 1; ++k) {
    X[0, k].set_value(
        std::lerp(x_initial[0], x_final[0], static_cast<double>(k) / N));
    X[1, k].set_value(
        std::lerp(x_initial[1], x_final[1], static_cast<double>(k) / N));
  }

  // u = f_x
  auto U = problem.decision_variable(1, N);

  problem.subject_to(X.col(0) == x_initial);
  problem.subject_to(X.col(N) == x_final);
  problem.subject_to(slp::bounds(0.0, X.row(0), d_max));
  problem.subject_to(slp::bounds(-u_max, U, u_max));

  for (int k = 0; k < N; ++k) {
    problem.subject_to(
        X.col(k + 1) ==
        rk4<decltype(cart_pole_dynamics), slp::VariableMatrix<double>,
            slp::VariableMatrix<double>>(cart_pole_dynamics, X.col(k), U.col(k),
                                         dt));
  }

  slp::Variable J = 0.0;
  for (int k = 0; k < N; ++k) {
    J += U.col(k).T() * U.col(k);
  }
  problem.minimize(J);

  return problem;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 8;
  constexpr std::chrono::duration<double> T{5.0};
  slp::Problem<double> problem = cart_pole_sleipnir(T / N, N);
  // problem.hpp:236-260; expected types of cart_pole_problem_test.cpp:87-89
  std::printf("n=%zu m_e=%zu m_i=%zu cost=%d eq=%d ineq=%d\n", problem.decision_variables().size(),
              problem.equality_constraints().size(), problem.inequality_constraints().size(),
              static_cast<int>(problem.cost_function_type()), static_cast<int>(problem.equality_constraint_type()),
              static_cast<int>(problem.inequality_constraint_type()));
  try {
    const slp::ExitStatus status = problem.solve(slp::Options{});
    std::printf("status=%d iterations=%d\n", static_cast<int>(status), problem.report().iterations);
    return status == slp::ExitStatus::SUCCESS ? 0 : 2;
  } catch (const std::exception& e) {
    // no HIP device: the product has no CPU fallback (it says so and stops)
    std::printf("solve: %s\n", e.what());
    return 3;
  }
}
