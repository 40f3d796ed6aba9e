// Drop-in include path of the reference (include/sleipnir/autodiff/expression_type.hpp): slp::ExpressionType lives with slp::Variable here.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/variable.hpp"
