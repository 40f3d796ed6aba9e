// Reference include path -> the MI355X-native surface (sleipnir_amd/csrc/slp/ocp.hpp).
#pragma once
#include "../../../sleipnir_amd/csrc/slp/ocp.hpp"
