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
#include <functional>
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

// What the flat compiler reports about every task it emitted, in terms of the caller's graph nodes, value outputs and
// rows: what the family code needs to instantiate the task for the other members of its family.
struct TapeTrace {
  struct Task {
    std::vector<NodeId> leaf_nodes;                     // graph node of every leaf, in the task's leaf order
    std::vector<uint32_t> vouts;                        // index into value_outs of every value output, in the task's order
    std::vector<std::pair<uint32_t, NodeId>> jouts;     // (index into rows, wrt node) of every derivative output
    uint32_t comp_nodes = 0;                            // interior nodes of its components (without private copies)
    uint32_t n_comps = 0;
    int cls = 0;
  };
  std::vector<Task> tasks;
};

// Families of structurally identical components (tape_compiler.cpp: "Families"), in three phases so that a caller can
// work on a family's REPRESENTATIVE between them (nlp.cpp builds the Hessian rows of a stage family on one stage only
// and never materializes the other stages' gradient expressions):
//   tape_families_analyze   components of the graph under (value_outs, rows), every component's reachable set and
//                           sequence, the classes of equal sequences;
//   tape_families_accept    one family: its representative — with `extra_rows` (indices into `rows`) the caller built
//                           on it — through the flat compiler alone; false: not a family the instances can share;
//   tape_families_emit      the program: what is in no accepted family through the flat compiler together, every
//                           accepted family's members instantiated from its representative by position.  A leaf of the
//                           representative's task that is NOT in its component's reachable set (a multiplier, a
//                           constant made by the caller) and the destination of an extra row's output are the caller's
//                           to translate for each member.
struct TapeFamilySet {
  size_t graph_size = 0;
  size_t ncomp = 0;
  std::vector<uint32_t> comp_start;       // interior members of component c: members[comp_start[c] .. comp_start[c + 1])
  std::vector<NodeId> members;            // ascending within a component
  std::vector<uint32_t> crow_start, crow, cvout_start, cvout;  // rows / value outputs of component c, in list order
  std::vector<uint32_t> loose_rows, loose_vouts;               // ... of no component (a bare leaf)
  std::vector<NodeId> all_nodes;          // every component's reachable set (members, private nodes, leaves), ascending
  std::vector<uint32_t> all_start;
  struct Family {
    uint32_t rep;                 // component
    std::vector<uint32_t> comps;  // its members, ascending (rep first)
  };
  std::vector<Family> fams;
  std::vector<NodeId> param_order;  // every parameter leaf the roots reach, in node order
  int32_t n_inputs = 0;
  struct Accepted {
    uint32_t fam;
    TapeProgram prog;
    TapeTrace::Task tt;
    std::vector<uint32_t> extra_rows;
  };
  std::vector<Accepted> accepted;
  std::vector<uint8_t> comp_in_family;  // per component: a member of an accepted family
};
struct TapeFamilyHooks {
  // the graph node member `comp` has where the representative of family `fam` has leaf `rep_leaf` (not of its component)
  std::function<NodeId(uint32_t fam, NodeId rep_leaf, uint32_t comp)> outside_leaf;
  // destination in V of output `out` of the family's extra row `k` (position in Accepted::extra_rows) for member `comp`
  std::function<int32_t(uint32_t fam, uint32_t k, uint32_t out, uint32_t comp)> extra_dst;
};
bool tape_families_analyze(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                           const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows, TapeFamilySet& S);
bool tape_families_accept(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                          const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows,
                          const TapeCompileOptions& opt, TapeFamilySet& S, uint32_t fam, const std::vector<uint32_t>& extra_rows);
// `more_vouts` / `more_rows`: what else goes through the flat compiler with the remainder (indices into the lists; rows
// and value outputs that were not part of the analysis)
TapeProgram tape_families_emit(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                               const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows,
                               const TapeCompileOptions& opt, TapeFamilySet& S, const TapeFamilyHooks& hooks,
                               const std::vector<uint32_t>& more_vouts = {}, const std::vector<uint32_t>& more_rows = {});

// `inputs`: leaf VAR node -> input vector index (nodes absent from the map must
// not be reachable).  Rows must reference wrt nodes that are in `inputs`.
TapeProgram compile_tape(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                         const std::vector<TapeValueOut>& value_outs,
                         const std::vector<TapeRow>& rows, const TapeCompileOptions& opt = {});

}  // namespace slpx
