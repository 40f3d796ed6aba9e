// Drop-in include path of the reference (include/sleipnir/autodiff/jacobian.hpp): Gradient, Jacobian and Hessian share one header here.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/derivatives.hpp"
