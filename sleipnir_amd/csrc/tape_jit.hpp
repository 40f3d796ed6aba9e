// Template kernels for the tape: one LANE per task instead of one workgroup per task.
//
// The stages of a transcribed optimal-control problem compile to tasks with byte-identical
// structure (tape_compiler.cpp shares it: "templates") — they differ only in which inputs
// they read and where their outputs go.  Interpreting such a task level by level inside a
// workgroup leaves 58 of 64 lanes idle and pays ~0.4 us of LDS round trips and interpreter
// instructions per level (75 + 76 levels for a cart-pole stage: 66 us).  The natural SIMT
// mapping is the other way round: the template becomes STRAIGHT-LINE code — every node a
// named double, every adjoint slot a chain of FMAs, register-allocated by the compiler — and
// lane i of the launch runs it on instance i's data.  The code is generated from the
// compiled TapeProgram and built at run time with hipRTC for gfx950; the arithmetic of
// every node is the same `op_forward` (tape_ops.h, embedded as the prelude of the generated
// source) the interpreting kernels call, in the same order.
//
// Tasks that are not part of a large enough template group (boundary stages, the cost row)
// stay on the interpreting kernels (tape_kernels.h).  Disabled with SLPX_TAPE_JIT=0.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "tape_compiler.hpp"

namespace slpx {

// One generated kernel and the tasks (indices into TapeProgram::tasks) it serves.
struct TapeTemplateGroup {
  std::vector<uint32_t> tasks;   // instances
  hipFunction_t fn = nullptr;    // extern "C" slpx_tape_template(...)
  uint32_t n_leaf = 0, n_node = 0, n_slot = 0;
  uint32_t n_groups = 1;         // row groups of the adjoint part (see tape_jit.cpp)
};

struct TapeJitResult {
  std::vector<TapeTemplateGroup> groups;
  std::vector<uint8_t> task_is_templated;  // per task of the program
  double compile_seconds = 0.0;
  std::string log;                         // non-empty when something fell back
};

// Finds the template groups of `prog` with at least `min_instances` members among the
// LDS-class tasks, generates + compiles (or fetches from the process-wide cache) one
// kernel per group.  Never throws: on any hipRTC problem the group is simply left to the
// interpreter and the reason is put in `log`.
TapeJitResult build_tape_templates(const TapeProgram& prog, uint32_t min_instances);

// Number of wave-uniform row groups the adjoint part of a template is split into.
uint32_t template_row_groups(const TapeProgram& prog, const TapeTask& representative);

// The generated source of one template (exposed for tests / inspection).
std::string generate_template_source(const TapeProgram& prog, const TapeTask& representative);

}  // namespace slpx
