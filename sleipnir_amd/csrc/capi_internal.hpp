// Definitions behind the opaque handles of include/slpx.h (shared with the
// test-only host checker in tests/support/).
#pragma once

#include <memory>

#include "newton.hpp"
#include "slp/problem.hpp"

struct slpx_system {
  std::unique_ptr<slpx::NewtonSystem> sys;
  slpx::NewtonSystem* ref = nullptr;
  // a handle from slpx_problem_system follows its problem: a model change (or a changed
  // parameter value) makes the problem compile a new system, and the handle resolves to that
  slp::Problem<double>* owner = nullptr;
  slpx::NewtonSystem& get() { return owner ? owner->compile() : *ref; }
};

struct slpx_problem {
  slp::Problem<double> problem;
  double t_compile = 0.0;
  std::unique_ptr<slpx_system> borrowed;  // slpx_problem_system()
};
