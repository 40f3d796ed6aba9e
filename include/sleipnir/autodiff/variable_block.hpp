// Drop-in include path of the reference (include/sleipnir/autodiff/variable_block.hpp): VariableBlock lives with the matrix classes here.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/variable.hpp"
