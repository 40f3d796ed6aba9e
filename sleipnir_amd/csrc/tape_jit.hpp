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
// ALL templates of a program live in ONE generated kernel (a wave-uniform switch on the
// block index picks the body), so the big family (the interior stages), the odd ones out
// (first / last stage) and the small fry (cost groups, packed linear rows) run side by side
// in a single launch instead of queueing behind each other.  Tasks left out (code-size cap,
// node-less tasks) stay on the interpreting kernels (tape_kernels.h).  SLPX_TAPE_JIT=0
// disables the whole mechanism.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "tape_compiler.hpp"

namespace slpx {

// One body of the generated kernel and the tasks (indices into TapeProgram::tasks) it serves.
struct TapeTemplateGroup {
  std::vector<uint32_t> tasks;   // instances
  uint32_t n_leaf = 0, n_node = 0, n_slot = 0;
  uint32_t n_groups = 1;         // row groups of the adjoint part (see tape_jit.cpp)
};

struct TapeJitResult {
  hipModule_t mod = nullptr;               // owner of fn (debug: slpx_tmpl_clocks lives in it)
  hipFunction_t fn = nullptr;              // extern "C" slpx_tape_templates(...), null = no templates
  std::vector<TapeTemplateGroup> groups;   // body k of the kernel serves groups[k]
  std::vector<uint8_t> task_is_templated;  // per task of the program
  std::vector<unsigned char> params;       // the table behind the kernel's last argument (TemplateParams::blob)
  bool specialized = false;                // the code object has this model's numbers as literals
  uint32_t block_threads = 64;             // threads of a workgroup of the kernel (256: the 256-thread interpreted tasks ride in it)
  double compile_seconds = 0.0;
  // TapeJitOptions::compile_without_device on a machine without a GPU: bodies that were
  // generated and compiled for gfx950 (>= 0), or -1 if hipRTC rejected the source
  int compiled_without_device = 0;
  std::string log;                         // non-empty when something fell back
};

struct TapeJitOptions {
  // A family needs this many members to get a body: one LANE runs the whole task, so a
  // lone task is far slower generated (one lane, serial: measured 229 us for ten packed
  // 320-node linear-row tasks) than interpreted by a workgroup (level-parallel, ~10 us).
  uint32_t min_instances = kTapeFamilyMin;
  // ... and so is a wide, shallow task (measured at N=5000: a family of packed linear-row
  // tasks, 640 loads + 320 stores per lane, turned a 45 us launch into 300 us)
  uint32_t max_width = 32;
  uint32_t max_bodies = 8;
  bool compile_without_device = false;  // tests: check that the generated source compiles
  uint32_t max_generated_nodes = 4000;  // code-size / compile-time cap over all bodies
  // inputs [0, n) whose scale factor (in_scale) is identically 1: the decision variables
  // (DeviceNlp::set_scaling scales only the multiplier inputs); leaves bound to them skip the factor
  uint32_t n_unscaled_inputs = 0;
  // 0: the kernel is the sweep; 1: it waits and signals through the `chain` words (DeviceNlp::sweep_full_for_step),
  // its stores to V written through to memory
  int chain_mode = 0;
  // non-empty: generate + compile WITHOUT a device and store the code object there (prebuild_tape_templates)
  std::string prebuild_dir;
};

// Groups the LDS-class tasks of `prog` by identical structure and output wiring, picks the
// groups worth a body, generates + compiles (or fetches from the process-wide cache) the
// kernel.  Never throws: on any hipRTC problem everything stays on the interpreter and the
// reason is put in `log`.
TapeJitResult build_tape_templates(const TapeProgram& prog, const TapeJitOptions& opt = {});

// Generates and compiles the kernel of `prog` for gfx950 without a device and stores the code
// object under `dir` (file name = hash of the generated source, as in the user cache), where
// build_tape_templates looks first (<libslpx.so dir>/jit_cache by default).  Returns the number
// of bodies, 0 if the program has no family worth one, -1 if hipRTC rejected the source.
int prebuild_tape_templates(const TapeProgram& prog, const TapeJitOptions& opt, const std::string& dir, std::string& log);

// The generated source for a set of families (members in instance order; exposed for tests /
// inspection).  The bindings of each body are specialized for its family, see tape_jit.cpp.
// `params` (optional): the model's numbers — base indices, constants — become members of a table
// in device memory the kernel's last argument points to (struct SlpxParams { double c[];
// unsigned w[]; }) instead of literals, and their values come back here: the source is then the
// same for every horizon and parameter value of a model family.  Without it, or when they exceed
// kTemplateParamBytesMax, they are literals and SlpxParams has one unused element of each kind.
struct TemplateParams {
  std::vector<uint32_t> w;
  std::vector<double> c;
  // the table as the kernel expects it: c[max(1, |c|)] then w[max(1, |w|)], padded to 8 bytes
  std::vector<unsigned char> blob() const;
};
constexpr size_t kTemplateParamBytesMax = 16384;  // beyond this the numbers stay literals
std::string generate_templates_source(const TapeProgram& prog, const std::vector<std::vector<uint32_t>>& families,
                                      uint32_t n_unscaled_inputs, TemplateParams* params = nullptr, uint32_t block_threads = 64);

// Number of wave-uniform row groups the adjoint part of a template is split into.
uint32_t template_row_groups(const TapeProgram& prog, const TapeTask& representative);

}  // namespace slpx
