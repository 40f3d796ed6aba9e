// Reference include path; the enumeration lives with slp::OCP (sleipnir_amd/csrc/slp/ocp.hpp).
#pragma once
#include "../ocp.hpp"
