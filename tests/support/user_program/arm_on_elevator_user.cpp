// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The arm-on-elevator problem of the reference's test
// (test/src/optimization/arm_on_elevator_problem_test.cpp:20-116): two double integrators over
// N = 800 steps of 5 ms, coupled by ONE nonlinear inequality per step (the end effector's height
// elevator + sin(arm angle) stays under 1.8 m, written with cwise_transform(slp::sin<T>)).  Same
// checks: QUADRATIC cost, LINEAR equalities, NONLINEAR inequalities, SUCCESS.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <numbers>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/optimization/problem.hpp>

int main(int argc, char** argv) {
  using T = double;
  const int N = argc > 1 ? std::atoi(argv[1]) : 800;

  constexpr T ELEVATOR_START_HEIGHT(1);      // m
  constexpr T ELEVATOR_END_HEIGHT(1.25);     // m
  constexpr T ELEVATOR_MAX_VELOCITY(1);      // m/s
  constexpr T ELEVATOR_MAX_ACCELERATION(2);  // m/s²

  constexpr T ARM_LENGTH(1);                                 // m
  constexpr T ARM_START_ANGLE(0);                            // rad
  constexpr T ARM_END_ANGLE(std::numbers::pi);               // rad
  constexpr T ARM_MAX_VELOCITY(2.0 * std::numbers::pi);      // rad/s
  constexpr T ARM_MAX_ACCELERATION(4.0 * std::numbers::pi);  // rad/s²

  constexpr T END_EFFECTOR_MAX_HEIGHT(1.8);  // m

  constexpr std::chrono::duration<T> TOTAL_TIME{T(4)};
  const auto dt = TOTAL_TIME / T(N);

  try {
    slp::Problem<T> problem;

    auto elevator = problem.decision_variable(2, N + 1);
    auto elevator_accel = problem.decision_variable(1, N);

    auto arm = problem.decision_variable(2, N + 1);
    auto arm_accel = problem.decision_variable(1, N);

    for (int k = 0; k < N; ++k) {
      // Elevator dynamics constraints
      problem.subject_to(elevator[0, k + 1] == elevator[0, k] + elevator[1, k] * dt.count() +
                                                   T(0.5) * elevator_accel[0, k] * dt.count() * dt.count());
      problem.subject_to(elevator[1, k + 1] == elevator[1, k] + elevator_accel[0, k] * dt.count());

      // Arm dynamics constraints
      problem.subject_to(arm[0, k + 1] ==
                         arm[0, k] + arm[1, k] * dt.count() + T(0.5) * arm_accel[0, k] * dt.count() * dt.count());
      problem.subject_to(arm[1, k + 1] == arm[1, k] + arm_accel[0, k] * dt.count());
    }

    // Elevator start and end conditions
    problem.subject_to(elevator.col(0) == slp::DenseMatrix{{ELEVATOR_START_HEIGHT}, {T(0)}});
    problem.subject_to(elevator.col(N) == slp::DenseMatrix{{ELEVATOR_END_HEIGHT}, {T(0)}});

    // Arm start and end conditions
    problem.subject_to(arm.col(0) == slp::DenseMatrix{{ARM_START_ANGLE}, {T(0)}});
    problem.subject_to(arm.col(N) == slp::DenseMatrix{{ARM_END_ANGLE}, {T(0)}});

    // Elevator velocity limits
    problem.subject_to(slp::bounds(-ELEVATOR_MAX_VELOCITY, elevator.row(1), ELEVATOR_MAX_VELOCITY));

    // Elevator acceleration limits
    problem.subject_to(slp::bounds(-ELEVATOR_MAX_ACCELERATION, elevator_accel, ELEVATOR_MAX_ACCELERATION));

    // Arm velocity limits
    problem.subject_to(slp::bounds(-ARM_MAX_VELOCITY, arm.row(1), ARM_MAX_VELOCITY));

    // Arm acceleration limits
    problem.subject_to(slp::bounds(-ARM_MAX_ACCELERATION, arm_accel, ARM_MAX_ACCELERATION));

    // Height limit
    auto heights = elevator.row(0) + ARM_LENGTH * arm.row(0).cwise_transform(slp::sin<T>);
    problem.subject_to(heights <= END_EFFECTOR_MAX_HEIGHT);

    // Cost function
    slp::Variable J = T(0);
    for (int k = 0; k < N + 1; ++k) {
      J += pow(ELEVATOR_END_HEIGHT - elevator[0, k], T(2)) + pow(ARM_END_ANGLE - arm[0, k], T(2));
    }
    problem.minimize(J);

    std::printf("cost=%d eq=%d ineq=%d\n", static_cast<int>(problem.cost_function_type()),
                static_cast<int>(problem.equality_constraint_type()),
                static_cast<int>(problem.inequality_constraint_type()));
    if (argc > 2) return 0;  // model only (no device needed)

    const auto status = problem.solve();
    // the constraints it was asked to respect, at the answer
    int bad = static_cast<int>(status) != 0;
    double top = 0.0;
    for (int k = 0; k < N + 1; ++k) top = std::fmax(top, elevator.value(0, k) + ARM_LENGTH * std::sin(arm.value(0, k)));
    bad += !(top <= END_EFFECTOR_MAX_HEIGHT + 1e-6);
    bad += !(std::abs(elevator.value(0, N) - ELEVATOR_END_HEIGHT) < 1e-6);
    bad += !(std::abs(arm.value(0, N) - ARM_END_ANGLE) < 1e-6);
    std::printf("status=%d highest end-effector point %.6f m, final (%.6f m, %.6f rad) failed_checks=%d\n",
                static_cast<int>(status), top, elevator.value(0, N), arm.value(0, N), bad);
    return bad == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
