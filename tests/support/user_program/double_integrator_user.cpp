// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The double-integrator minimum-error problem of the reference's test
// (test/src/optimization/double_integrator_problem_test.cpp:24-134): 700 steps of 5 ms, velocity
// and acceleration limits of 1.  Same checks: QUADRATIC cost, LINEAR equalities and inequalities,
// SUCCESS, the bang-coast-bang input profile to 1e-4 (transitions anywhere inside the limits),
// states within 1e-2 of the discrete model driven by that profile, end points to 1e-8.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <exception>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/optimization/problem.hpp>

int main(int argc, char**) {
  using T = double;
  using std::abs;

  constexpr std::chrono::duration<T> TOTAL_TIME{T(3.5)};
  constexpr std::chrono::duration<T> dt{T(0.005)};
  constexpr int N = static_cast<int>(TOTAL_TIME / dt);

  constexpr T r(2);  // m

  try {
    slp::Problem<T> problem;

    // 2x1 state vector with N + 1 timesteps (includes last state)
    auto X = problem.decision_variable(2, N + 1);

    // 1x1 input vector with N timesteps (input at last state doesn't matter)
    auto U = problem.decision_variable(1, N);

    // Kinematics constraint assuming constant acceleration between timesteps
    for (int k = 0; k < N; ++k) {
      constexpr T t = dt.count();
      auto p_k1 = X[0, k + 1];
      auto v_k1 = X[1, k + 1];
      auto p_k = X[0, k];
      auto v_k = X[1, k];
      auto a_k = U[0, k];

      // pₖ₊₁ = pₖ + vₖt + 1/2aₖt²
      problem.subject_to(p_k1 == p_k + v_k * t + 0.5 * a_k * t * t);

      // vₖ₊₁ = vₖ + aₖt
      problem.subject_to(v_k1 == v_k + a_k * t);
    }

    // Start and end at rest
    problem.subject_to(X.col(0) == slp::DenseMatrix{{T(0)}, {T(0)}});
    problem.subject_to(X.col(N) == slp::DenseMatrix{{r}, {T(0)}});

    // Limit velocity
    problem.subject_to(slp::bounds(T(-1), X.row(1), T(1)));

    // Limit acceleration
    problem.subject_to(slp::bounds(T(-1), U, T(1)));

    // Cost function - minimize position error
    slp::Variable J = T(0);
    for (int k = 0; k < N + 1; ++k) {
      J += pow(r - X[0, k], 2);
    }
    problem.minimize(J);

    std::printf("cost=%d eq=%d ineq=%d\n", static_cast<int>(problem.cost_function_type()),
                static_cast<int>(problem.equality_constraint_type()),
                static_cast<int>(problem.inequality_constraint_type()));
    if (argc > 1) return 0;  // model only (no device needed)

    const auto status = problem.solve();
    int bad = static_cast<int>(status) != 0;

    // x ← A x + B u with A = [1 dt; 0 1], B = [dt²/2; dt]
    T x0 = 0, x1 = 0, u = 0;
    bad += !(abs(X.value(0, 0)) < 1e-8) + !(abs(X.value(1, 0)) < 1e-8);
    for (int k = 0; k < N; ++k) {
      bad += !(abs(X.value(0, k) - x0) < 1e-2);
      bad += !(abs(X.value(1, k) - x1) < 1e-2);

      // expected input for this timestep
      if (T(k) * dt < std::chrono::duration<T>{T(1)}) u = T(1);            // accelerate
      else if (T(k) * dt < std::chrono::duration<T>{T(2.05)}) u = T(0);    // maintain speed
      else if (T(k) * dt < std::chrono::duration<T>{T(3.275)}) u = T(-1);  // decelerate
      else u = T(1);                                                       // accelerate

      if (k > 0 && k < N - 1 && abs(U.value(0, k - 1) - U.value(0, k + 1)) >= T(1.0 - 1e-2)) {
        // transitioning between -1, 0 and 1: anywhere within the limits
        bad += !(U.value(0, k) >= T(-1)) + !(U.value(0, k) <= T(1));
      } else {
        bad += !(abs(U.value(0, k) - u) < 1e-4);
      }

      // project the state forward
      const T n0 = x0 + dt.count() * x1 + T(0.5) * dt.count() * dt.count() * u;
      x1 = x1 + dt.count() * u;
      x0 = n0;
    }
    bad += !(abs(X.value(0, N) - r) < 1e-8) + !(abs(X.value(1, N)) < 1e-8);
    std::printf("status=%d final=(%.9f, %.2e) failed_checks=%d\n", static_cast<int>(status), X.value(0, N),
                X.value(1, N), bad);
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
