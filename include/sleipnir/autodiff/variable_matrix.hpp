// Drop-in include path of the reference (include/sleipnir/autodiff/variable_matrix.hpp):
// Variable, VariableMatrix and VariableBlock live in one header here.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/variable.hpp"
