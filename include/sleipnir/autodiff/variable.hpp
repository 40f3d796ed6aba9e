// Drop-in include path of the reference (include/sleipnir/autodiff/variable.hpp).
#pragma once
#include "../../../sleipnir_amd/csrc/slp/variable.hpp"
