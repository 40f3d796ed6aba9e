// Tape compiler: turns the host expression graph plus a list of "rows" (roots
// whose sparse gradients are wanted) into the flat, partitioned, levelized
// program the `tape_sweep` HIP kernel executes.
//
// What it replaces in the reference (all re-done per call on the CPU there):
//   update_values   expression_graph.hpp:86-96   (once per ROW list, jacobian.hpp:139-141)
//   append_triplets expression_graph.hpp:107-153 (once per nonlinear row, jacobian.hpp:148-151)
//   setFromTriplets jacobian.hpp:153 / hessian.hpp:151 (sort + compress per call)
//
// MI355X design:
//   * the union of all rows' graphs is evaluated ONCE per sweep (shared nodes are
//     not re-evaluated per row);
//   * the graph is split into connected components of interior nodes (stages of
//     a direct-transcription problem fall out as components because they only
//     share leaves); components are bin-packed into TASKS whose working set
//     (values, local partials, adjoint slots) fits in LDS; one workgroup runs a
//     task start to finish, so there is no grid-wide synchronisation at all;
//   * the per-row reverse sweep becomes a levelized GATHER over (row,node)
//     adjoint slots: slot = sum over parent slots of parent_adjoint * partial.
//     Edges of a slot are stored in the row's parent->child list order, i.e.
//     the order the reference accumulates `child.adjoint +=`, so sums associate
//     identically;
//   * outputs are written straight into fixed positions of one value vector V
//     (static sparsity patterns; no triplets, no sorting).
#pragma once

#include <cstdint>
#include <utility>
#include <vector>

#include "graph.hpp"
#include "tape_device.h"

namespace slpx {

struct TapeRow {
  NodeId root = kNull;
  // (wrt node, destination index in V, scale index) for each structural nonzero
  struct Out {
    NodeId wrt;
    int32_t dst;
  };
  std::vector<Out> outputs;
  int32_t scale_idx = -1;  // index into the scale vector, -1 = unscaled
};

struct TapeValueOut {
  NodeId node;
  int32_t dst;
  int32_t scale_idx;
};

struct TapeProgram {
  int32_t n_inputs = 0;
  int32_t n_outputs = 0;
  std::vector<TapeTask> tasks;
  // task classes: indices into `tasks`
  std::vector<uint32_t> small_tasks, large_tasks, global_tasks;
  std::vector<uint32_t> leaf_src;
  std::vector<double> consts;
  // Parameters: free Variables that are not tape inputs.  Each owns one slot of
  // `consts` (never shared with a literal), so its value can be refreshed on the device
  // without recompiling: (graph node, index into consts).
  std::vector<std::pair<NodeId, uint32_t>> params;
  uint32_t shared_tasks = 0;  // tasks whose structure (records, levels, edges) is another task's
  std::vector<uint32_t> node_rec;  // [op | need_dl<<8 | need_dr<<9, a0, a1] per node
  std::vector<uint32_t> lvl_ptr;
  std::vector<uint32_t> slot_edge_ptr;
  std::vector<uint32_t> slvl_ptr;
  std::vector<TapeEdge> edges;
  // 16-bit packed mirror of node_rec / slot_edge_ptr / edges, indexed with the same
  // task offsets; what the LDS-staged kernel loads (GLOBAL tasks use the 32-bit arrays)
  std::vector<uint16_t> node_rec16;       // [op|flags, a0, a1, 0] per node
  std::vector<uint16_t> slot_edge_ptr16;
  std::vector<uint16_t> edges16;          // [parent_slot, partial] per edge
  std::vector<uint32_t> vout_src;    // local value index
  std::vector<uint32_t> vout_dst;
  std::vector<int32_t> vout_scale;
  std::vector<uint32_t> jout_slot;   // local slot index
  std::vector<uint32_t> jout_dst;
  std::vector<int32_t> jout_scale;
  uint32_t small_lds_bytes = 0, large_lds_bytes = 0;
  uint64_t global_scratch_doubles = 0;
  bool basic_ops = true;  // every op is in the basic set (tape_ops.h op_is_basic)

  // statistics
  size_t total_nodes = 0, total_slots = 0, total_edges = 0, total_leaves = 0;
  uint32_t max_levels = 0, max_slot_levels = 0;
};

// Structurally identical components from this many up are a FAMILY: each keeps a task of its own
// (tape_compiler.cpp) and they run as instances of one generated body, a lane each (tape_jit.cpp).
constexpr uint32_t kTapeFamilyMin = 8;

struct TapeCompileOptions {
  bool cse = true;                        // merge structurally identical interior nodes (SLPX_TAPE_CSE=0: off)
  // compile ONE member of every family of structurally identical components and instantiate the others from it
  // (compile_tape_families; SLPX_TAPE_TEMPLATES=0: every component through the flat compiler, the same program)
  bool families = true;
  uint32_t small_lds_bytes = 40 * 1024;   // 64-thread workgroups, four per CU
  uint32_t large_lds_bytes = 152 * 1024;  // 256-thread workgroups, one per CU
  bool rebalance_sums = true;
  uint32_t rebalance_min_terms = 8;
  // separable cost sums (nlp.cpp): cut into groups of `split_sum_group` terms when the sum
  // has at least `split_sum_min_terms` terms (0 = never)
  uint32_t split_sum_min_terms = 64;
  uint32_t split_sum_group = 32;
};

// `inputs`: leaf VAR node -> input vector index (nodes absent from the map must
// not be reachable).  Rows must reference wrt nodes that are in `inputs`.
TapeProgram compile_tape(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                         const std::vector<TapeValueOut>& value_outs,
                         const std::vector<TapeRow>& rows, const TapeCompileOptions& opt = {});

}  // namespace slpx
