// Benchmark model builders (see problems.cpp).
#pragma once

#include "slp/problem.hpp"

namespace slpx_models {

void build_cart_pole(slp::Problem& problem, double dt, int N, slp::VariableMatrix* X = nullptr,
                     slp::VariableMatrix* U = nullptr);
void build_flywheel(slp::Problem& problem, double dt, int N, slp::VariableMatrix* X = nullptr,
                    slp::VariableMatrix* U = nullptr);

}  // namespace slpx_models
