// Drop-in include path of the reference (include/sleipnir/optimization/multistart.hpp): slp::multistart.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/multistart.hpp"
