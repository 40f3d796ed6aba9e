// Drop-in include path of the reference (include/sleipnir/optimization/problem.hpp): a user
// program keeps `#include <sleipnir/optimization/problem.hpp>` and gets slp::Problem<double>
// backed by libslpx (sleipnir_amd/csrc/slp/problem.hpp).  Build with -I<repo>/include.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/problem.hpp"
