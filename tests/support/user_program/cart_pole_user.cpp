// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The cart-pole model of the reference's scalability benchmark
// (benchmarks/scalability/cart_pole/sleipnir.cpp:16-129, benchmarks/rk4.hpp:14-23), written
// against the reference's own include lines and with every spelling of its API that program
// uses — slp::VariableMatrix<double>, slp::Problem<double>, problem.decision_variable(4, N + 1),
// X[0, k].set_value, X.col(k), x.segment(0, 2), slp::bounds, solve(M, ...), U.col(k).T() * U.col(k),
// slp::Variable J = 0.0, rk4<decltype(f), slp::VariableMatrix<double>, slp::VariableMatrix<double>> —
// compiled against <repo>/include and linked with libslpx.so by tests/test_slp_surface.py.
// Eigen::Matrix / Eigen::Vector constants are slp::DenseMatrix (Eigen is not in this toolchain).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <numbers>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/problem.hpp>

// classical Runge-Kutta step with the benchmark's template parameter list (benchmarks/rk4.hpp:14-23)
template <typename F, typename T, typename U>
T rk4(F&& f, T x, U u, std::chrono::duration<double> dt) {
  const double h = dt.count(), half = 0.5 * h;
  const T s1 = f(x, u);
  const T s2 = f(x + half * s1, u);
  const T s3 = f(x + half * s2, u);
  const T s4 = f(x + h * s3, u);
  return x + (h / 6.0) * (s1 + 2.0 * s2 + 2.0 * s3 + s4);
}

// M(q) q'' + C(q, q') q' = tau_g(q) + B u for a pole on a cart; x = [q, q'] with q = [position, angle]
slp::VariableMatrix<double> cart_pole_dynamics(const slp::VariableMatrix<double>& x, const slp::VariableMatrix<double>& u) {
  constexpr double cart_mass = 5.0, pole_mass = 0.5, pole_length = 0.5, gravity = 9.806;  // kg, kg, m, m/s²
  constexpr double ml = pole_mass * pole_length;

  auto q = x.segment(0, 2);
  auto qdot = x.segment(2, 2);
  auto c = cos(q[1]);
  auto s = sin(q[1]);

  slp::VariableMatrix<double> M{{cart_mass + pole_mass, ml * c}, {ml * c, ml * pole_length}};
  slp::VariableMatrix<double> C{{0, -ml * qdot[1] * s}, {0, 0}};
  slp::VariableMatrix<double> tau_g{{0}, {-ml * gravity * s}};
  slp::DenseMatrix B{{1}, {0}};  // (an Eigen::Matrix<double, 2, 1> in the benchmark)

  slp::VariableMatrix<double> xdot{4, 1};
  xdot.segment(0, 2) = qdot;
  xdot.segment(2, 2) = solve(M, tau_g - C * qdot + B * u);
  return xdot;
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

  // straight-line initial guess for the cart position and the pole angle
  for (int k = 0; k <= N; ++k) {
    const double t = static_cast<double>(k) / N;
    X[0, k].set_value(std::lerp(x_initial[0], x_final[0], t));
    X[1, k].set_value(std::lerp(x_initial[1], x_final[1], t));
  }

  // u = f_x
  auto U = problem.decision_variable(1, N);

  problem.subject_to(X.col(0) == x_initial);
  problem.subject_to(X.col(N) == x_final);
  problem.subject_to(slp::bounds(0.0, X.row(0), d_max));
  problem.subject_to(slp::bounds(-u_max, U, u_max));

  // dynamics rows and the cost (sum of squared inputs) in one pass over the steps
  slp::Variable J = 0.0;
  for (int k = 0; k < N; ++k) {
    using Mat = slp::VariableMatrix<double>;
    problem.subject_to(X.col(k + 1) == rk4<decltype(cart_pole_dynamics), Mat, Mat>(cart_pole_dynamics, X.col(k), U.col(k), dt));
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
