// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The differential-drive problem of the reference's test
// (test/src/optimization/differential_drive_problem_test.cpp:26-119 with the model of
// test/include/differential_drive_util.hpp and test/include/rk4.hpp): 100 steps of 50 ms, RK4
// dynamics constraints, drive from the origin to (1, 1) at rest within ±12 V, minimum sum of
// squared states and inputs.  Same checks: QUADRATIC cost, NONLINEAR equalities, LINEAR
// inequalities, SUCCESS, every state to 1e-8 of the model integrated with the inputs found.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <exception>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/problem.hpp>

namespace {
using T = double;
constexpr T trackwidth = 0.699, Kv_linear = 3.02, Ka_linear = 0.642, Kv_angular = 1.382, Ka_angular = 0.08495;
constexpr T A1 = -(Kv_linear / Ka_linear + Kv_angular / Ka_angular) / 2.0;
constexpr T A2 = -(Kv_linear / Ka_linear - Kv_angular / Ka_angular) / 2.0;
constexpr T B1 = 0.5 / Ka_linear + 0.5 / Ka_angular;
constexpr T B2 = 0.5 / Ka_linear - 0.5 / Ka_angular;

template <typename F, typename State, typename Input>
State rk4(F&& f, State x, Input u, std::chrono::duration<T> dt) {
  const auto h = dt.count();
  State k1 = f(x, u);
  State k2 = f(x + h * 0.5 * k1, u);
  State k3 = f(x + h * 0.5 * k2, u);
  State k4 = f(x + h * k3, u);
  return x + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
}

// x = [x, y, heading, left velocity, right velocity], u = [left voltage, right voltage]
slp::VariableMatrix<T> dynamics_variable(const slp::VariableMatrix<T>& x, const slp::VariableMatrix<T>& u) {
  slp::VariableMatrix<T> xdot{5};
  auto v = (x[3] + x[4]) / T(2);
  xdot[0] = v * cos(x[2]);
  xdot[1] = v * sin(x[2]);
  xdot[2] = (x[4] - x[3]) / trackwidth;
  slp::DenseMatrix A{{A1, A2}, {A2, A1}};
  slp::DenseMatrix B{{B1, B2}, {B2, B1}};
  xdot.segment(3, 2) = A * x.segment(3, 2) + B * u;
  return xdot;
}

// the same model on plain numbers
struct State5 {
  T v[5];
};
State5 operator+(const State5& a, const State5& b) {
  State5 r;
  for (int i = 0; i < 5; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
State5 operator*(T s, const State5& a) {
  State5 r;
  for (int i = 0; i < 5; ++i) r.v[i] = s * a.v[i];
  return r;
}
struct Input2 {
  T v[2];
};
State5 dynamics_scalar(const State5& x, const Input2& u) {
  State5 d;
  const T v = (x.v[3] + x.v[4]) / T(2);
  d.v[0] = v * std::cos(x.v[2]);
  d.v[1] = v * std::sin(x.v[2]);
  d.v[2] = (x.v[4] - x.v[3]) / trackwidth;
  d.v[3] = A1 * x.v[3] + A2 * x.v[4] + B1 * u.v[0] + B2 * u.v[1];
  d.v[4] = A2 * x.v[3] + A1 * x.v[4] + B2 * u.v[0] + B1 * u.v[1];
  return d;
}
T lerp(T a, T b, T t) { return a + (b - a) * t; }
}  // namespace

int main(int argc, char**) {
  constexpr std::chrono::duration<T> TOTAL_TIME{T(5)};
  constexpr std::chrono::duration<T> dt{T(0.05)};
  constexpr int N = static_cast<int>(TOTAL_TIME / dt);

  constexpr T u_max(12);  // V

  const slp::DenseMatrix x_initial{{T(0)}, {T(0)}, {T(0)}, {T(0)}, {T(0)}};
  const slp::DenseMatrix x_final{{T(1)}, {T(1)}, {T(0)}, {T(0)}, {T(0)}};

  try {
    slp::Problem<T> problem;

    // x = [x, y, heading, left velocity, right velocity]ᵀ
    auto X = problem.decision_variable(5, N + 1);

    // Initial guess
    for (int k = 0; k < N; ++k) {
      X[0, k].set_value(lerp(x_initial[0], x_final[0], T(k) / T(N)));
      X[1, k].set_value(lerp(x_initial[1], x_final[1], T(k) / T(N)));
    }

    // u = [left voltage, right voltage]ᵀ
    auto U = problem.decision_variable(2, N);

    // Initial conditions
    problem.subject_to(X.col(0) == x_initial);

    // Final conditions
    problem.subject_to(X.col(N) == x_final);

    // Input constraints
    problem.subject_to(slp::bounds(-u_max, U, u_max));

    // Dynamics constraints - RK4 integration
    for (int k = 0; k < N; ++k) {
      problem.subject_to(X.col(k + 1) ==
                         rk4<decltype(dynamics_variable), slp::VariableMatrix<T>, slp::VariableMatrix<T>>(
                             dynamics_variable, X.col(k), U.col(k), dt));
    }

    // Minimize sum squared states and inputs
    slp::Variable J = T(0);
    for (int k = 0; k < N; ++k) {
      J += X.col(k).T() * X.col(k) + U.col(k).T() * U.col(k);
    }
    problem.minimize(J);

    std::printf("cost=%d eq=%d ineq=%d\n", static_cast<int>(problem.cost_function_type()),
                static_cast<int>(problem.equality_constraint_type()),
                static_cast<int>(problem.inequality_constraint_type()));
    if (argc > 1) return 0;  // model only (no device needed)

    const auto status = problem.solve();
    int bad = static_cast<int>(status) != 0;

    for (int r = 0; r < 5; ++r) bad += !(std::abs(X.value(r, 0) - x_initial[r]) < 1e-8);
    State5 x{{0, 0, 0, 0, 0}};
    double worst = 0.0;
    for (int k = 0; k < N; ++k) {
      const slp::DenseMatrix uk = U.col(k).value();
      const Input2 u{{uk[0], uk[1]}};
      for (int r = 0; r < 2; ++r) bad += !(U[r, k].value() >= -u_max) + !(U[r, k].value() <= u_max);
      for (int r = 0; r < 5; ++r) {
        worst = std::fmax(worst, std::abs(X.value(r, k) - x.v[r]));
        bad += !(std::abs(X.value(r, k) - x.v[r]) < 1e-8);
      }
      x = rk4(dynamics_scalar, x, u, dt);
    }
    for (int r = 0; r < 5; ++r) bad += !(std::abs(X.value(r, N) - x_final[r]) < 1e-8);
    std::printf("status=%d final=(%.9f, %.9f, %.2e) worst state error %.2e failed_checks=%d\n",
                static_cast<int>(status), X.value(0, N), X.value(1, N), X.value(2, N), worst, bad);
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
