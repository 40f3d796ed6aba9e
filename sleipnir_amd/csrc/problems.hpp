// Benchmark model builders (see problems.cpp).
#pragma once

#include "slp/problem.hpp"

namespace slpx_models {

void build_cart_pole(slp::Problem<double>& problem, double dt, int N, slp::VariableMatrix<double>* X = nullptr,
                     slp::VariableMatrix<double>* U = nullptr);
void build_flywheel(slp::Problem<double>& problem, double dt, int N, slp::VariableMatrix<double>* X = nullptr,
                    slp::VariableMatrix<double>* U = nullptr);

}  // namespace slpx_models
