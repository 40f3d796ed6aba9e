// Drop-in include path of the reference (include/sleipnir/optimization/solver/options.hpp): slp::Options lives with slp::Problem here.
#pragma once
#include "../../../../sleipnir_amd/csrc/slp/problem.hpp"
