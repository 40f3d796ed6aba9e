// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// slp::multistart on Mishra's bird function, the reference's test
// (test/src/optimization/multistart_test.cpp:16-53): two starts on two threads, each building
// and solving its own problem; the better optimum (-3.13024680, -1.58214218) to 1e-8.
#include <cmath>
#include <cstdio>
#include <exception>
#include <vector>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/optimization/multistart.hpp>
#include <sleipnir/optimization/problem.hpp>
#include <sleipnir/optimization/solver/exit_status.hpp>

int main() {
  using T = double;
  struct DecisionVariables {
    T x;
    T y;
  };
  try {
    auto solve = [](const DecisionVariables& input) -> slp::MultistartResult<T, DecisionVariables> {
      slp::Problem<T> problem;

      auto x = problem.decision_variable();
      auto y = problem.decision_variable();
      x.set_value(input.x);
      y.set_value(input.y);

      slp::Variable J = sin(y) * exp(pow(T(1) - cos(x), T(2))) + cos(x) * exp(pow(T(1) - sin(y), T(2))) + pow(x - y, T(2));
      problem.minimize(J);

      problem.subject_to(pow(x + T(5), T(2)) + pow(y + T(5), T(2)) < T(25));

      return {problem.solve(), J.value(), DecisionVariables{x.value(), y.value()}};
    };

    auto [status, cost, variables] = slp::multistart<T, DecisionVariables>(
        solve, std::vector{DecisionVariables{T(-3), T(-8)}, DecisionVariables{T(-3), T(-1.5)}});

    int bad = status != slp::ExitStatus::SUCCESS;
    bad += !(std::abs(variables.x - T(-3.13024680)) < 1e-8) + !(std::abs(variables.y - T(-1.58214218)) < 1e-8);
    std::printf("status=%d cost=%.9f at (%.9f, %.9f) failed_checks=%d\n", static_cast<int>(status), cost, variables.x,
                variables.y, bad);
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
