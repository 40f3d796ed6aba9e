// TEST / BENCH FIXTURE — tests/support/libslpx_models.so: the benchmark models (bench_models.hpp)
// built as user programs of the slp:: surface and handed to the product as slpx_problem handles
// through the public C-ABI only.  bench.py, the tests and the profiling scripts get their models
// here; the product library contains none.
#include "bench_models.hpp"

#include <exception>
#include <string>

#include <slpx.h>

namespace {
thread_local std::string g_error;

template <typename Build>
slpx_problem* as_handle(Build&& build) {
  try {
    slp::Problem<double> model;
    build(model);
    slpx_problem* p = slpx_problem_create();
    for (const auto& v : model.decision_variables()) slpx_problem_adopt_variable(p, v.expr);
    if (model.cost_function_type() != slp::ExpressionType::NONE) slpx_problem_minimize(p, model.cost().expr);
    for (const auto& c : model.equality_constraints()) slpx_problem_subject_to_eq(p, c.expr);
    for (const auto& c : model.inequality_constraints()) slpx_problem_subject_to_ineq(p, c.expr);
    return p;
  } catch (const std::exception& e) {
    g_error = e.what();
    return nullptr;
  }
}
}  // namespace

extern "C" {
const char* bench_models_last_error(void) { return g_error.c_str(); }
slpx_problem* bench_models_cart_pole(int32_t N, double dt) {
  return as_handle([&](slp::Problem<double>& m) { bench_models::build_cart_pole(m, dt, N); });
}
slpx_problem* bench_models_flywheel(int32_t N, double dt) {
  return as_handle([&](slp::Problem<double>& m) { bench_models::build_flywheel(m, dt, N); });
}
}
