// Drop-in include path of the reference (include/sleipnir/autodiff/slice.hpp): slp::Slice and slp::slicing::_ live with the matrix classes here.
#pragma once
#include "../../../sleipnir_amd/csrc/slp/variable.hpp"
