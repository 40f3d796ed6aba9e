// slp::multistart (include/sleipnir/optimization/multistart.hpp:17-79): the same problem solved
// from several initial guesses at once, the best result returned — successful solves before
// unsuccessful ones, lower cost first among equals.
//
// One thread per guess like the reference.  The expression graph of libslpx is per thread, so
// the user's `solve` builds its slp::Problem inside the call (the reference's own test does,
// multistart_test.cpp:24-43); every solve compiles its own system and runs on its own stream of
// the device.  For MANY starts of one model the batch interface is the better tool
// (slpx_system_create(batch = starts): one compiled system, all starts per launch — INTEGRATION.md §5).
#pragma once

#include <algorithm>
#include <functional>
#include <future>
#include <span>
#include <vector>

#include "problem.hpp"

namespace slp {

template <typename Scalar, typename DecisionVariables>
struct MultistartResult {
  ExitStatus status;
  Scalar cost;
  DecisionVariables variables;
};

template <typename Scalar, typename DecisionVariables>
MultistartResult<Scalar, DecisionVariables> multistart(
    const std::function<MultistartResult<Scalar, DecisionVariables>(const DecisionVariables& initial_guess)>& solve,
    std::span<const DecisionVariables> initial_guesses) {
  using Result = MultistartResult<Scalar, DecisionVariables>;
  std::vector<std::future<Result>> running;
  running.reserve(initial_guesses.size());
  for (const DecisionVariables& guess : initial_guesses)
    running.push_back(std::async(std::launch::async, [&solve, &guess] { return solve(guess); }));
  std::vector<Result> results;
  results.reserve(running.size());
  for (auto& r : running) results.push_back(r.get());
  return *std::min_element(results.begin(), results.end(), [](const Result& a, const Result& b) {
    const bool a_ok = a.status == ExitStatus::SUCCESS, b_ok = b.status == ExitStatus::SUCCESS;
    return a_ok != b_ok ? a_ok : a.cost < b.cost;
  });
}

}  // namespace slp
