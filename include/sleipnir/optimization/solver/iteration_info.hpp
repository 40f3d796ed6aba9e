// Drop-in include path of the reference (include/sleipnir/optimization/solver/iteration_info.hpp): slp::IterationInfo lives with slp::Problem here.
#pragma once
#include "../../../../sleipnir_amd/csrc/slp/problem.hpp"
