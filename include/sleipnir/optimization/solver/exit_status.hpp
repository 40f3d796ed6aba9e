// Drop-in include path of the reference (include/sleipnir/optimization/solver/exit_status.hpp): slp::ExitStatus lives with slp::Problem here.
#pragma once
#include "../../../../sleipnir_amd/csrc/slp/problem.hpp"
