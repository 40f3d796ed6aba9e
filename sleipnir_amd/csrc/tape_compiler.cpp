#include "tape_compiler.hpp"

#include <cstdlib>

#include "setup_timing.hpp"
#include "setup_threads.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <future>
#include <memory>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <unordered_map>

namespace slpx {

namespace {

// Compile-time working copy of the reachable part of the graph.
struct CG {
  std::vector<uint8_t> op;
  std::vector<int32_t> a0, a1;     // CG ids, -1 = none
  std::vector<NodeId> src;         // originating graph node (-1 for synthesized)
  std::vector<int32_t> scratch;    // in-degree counter for the reference ordering
  int32_t add(uint8_t o, int32_t l, int32_t r, NodeId s) {
    op.push_back(o);
    a0.push_back(l);
    a1.push_back(r);
    src.push_back(s);
    scratch.push_back(-1);
    return static_cast<int32_t>(op.size()) - 1;
  }
  size_t size() const { return op.size(); }
  bool is_leaf(int32_t n) const { return a0[n] < 0; }
};

// expression_graph.hpp:29-78 on the working copy
std::vector<int32_t> reference_order(CG& cg, int32_t root) {
  std::vector<int32_t> list;
  std::vector<int32_t> stack{root};
  while (!stack.empty()) {
    int32_t n = stack.back();
    stack.pop_back();
    for (int32_t arg : {cg.a0[n], cg.a1[n]})
      if (arg >= 0 && ++cg.scratch[arg] == 0) stack.push_back(arg);
  }
  stack.push_back(root);
  while (!stack.empty()) {
    int32_t n = stack.back();
    stack.pop_back();
    list.push_back(n);
    for (int32_t arg : {cg.a0[n], cg.a1[n]})
      if (arg >= 0 && --cg.scratch[arg] == -1) stack.push_back(arg);
  }
  return list;
}

struct UnionFind {
  std::vector<int32_t> p;
  explicit UnionFind(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0); }
  int32_t find(int32_t x) {
    while (p[x] != x) {
      p[x] = p[p[x]];
      x = p[x];
    }
    return x;
  }
  void unite(int32_t a, int32_t b) {
    a = find(a);
    b = find(b);
    if (a != b) p[std::max(a, b)] = std::min(a, b);
  }
};

}  // namespace

// the staged kernel reads whole 16-byte groups: every array it stages ends in 16 spare elements
static void add_tail_padding(TapeProgram& prog) {
  for (int k = 0; k < 16; ++k) {
    prog.node_rec16.push_back(0);
    prog.edges16.push_back(0);
    prog.slot_edge_ptr16.push_back(0);
    prog.lvl_ptr.push_back(0);
    prog.slvl_ptr.push_back(0);
    prog.leaf_src.push_back(0);
  }
}

// Per-graph-node tables of the flat compiler, kept between its calls by a caller that makes many (one per family
// representative: a call on a hundred nodes paid for tables of a million): `seen` and `to_cg` are returned to their
// blank state by the call that used them.
struct FlatScratch {
  std::vector<uint8_t> seen;      // reachable from the call's roots
  std::vector<int32_t> to_cg;     // graph node -> working-copy node, -1
  std::vector<int32_t> input_idx; // graph node -> tape input, -1
  int32_t n_inputs = 0;
  const std::vector<std::pair<NodeId, int32_t>>* bound = nullptr;
  FlatScratch() = default;
  FlatScratch(size_t G, const std::vector<std::pair<NodeId, int32_t>>& inputs) { bind(G, inputs); }
  // (the tables only grow; a thread that compiles model after model keeps them — and their pages — for good)
  void bind(size_t G, const std::vector<std::pair<NodeId, int32_t>>& inputs) {
    unbind();
    if (seen.size() < G) {
      seen.resize(G, 0);
      to_cg.resize(G, -1);
      input_idx.resize(G, -1);
    }
    n_inputs = 0;
    for (auto& [node, idx] : inputs) {
      input_idx[node] = idx;
      n_inputs = std::max(n_inputs, idx + 1);
    }
    bound = &inputs;
  }
  void unbind() {
    if (bound != nullptr)
      for (auto& [node, idx] : *bound) input_idx[node] = -1;
    bound = nullptr;
  }
};

// The flat compiler: everything reachable from the selected value outputs (`vsel`: indices into value_outs) and rows
// (`rsel`) as one program.  `tail_padding`: the 16 trailing elements the staged kernel's 16-byte loads may touch.
static TapeProgram compile_tape_flat(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                                     const std::vector<TapeValueOut>& value_outs_all, const std::vector<TapeRow>& rows_all,
                                     const std::vector<uint32_t>& vsel, const std::vector<uint32_t>& rsel,
                                     const TapeCompileOptions& opt, TapeTrace* trace, bool tail_padding,
                                     const std::vector<NodeId>* param_order = nullptr, FlatScratch* scratch = nullptr) {
  TapeProgram prog;
  std::unique_ptr<FlatScratch> own_scratch;
  if (scratch == nullptr) {
    own_scratch = std::make_unique<FlatScratch>(g.size(), inputs);
    scratch = own_scratch.get();
  }
  // (the selection, addressed like the whole lists were)
  struct ValueOutView {
    const std::vector<TapeValueOut>& all;
    const std::vector<uint32_t>& sel;
    size_t size() const { return sel.size(); }
    const TapeValueOut& operator[](size_t i) const { return all[sel[i]]; }
  } value_outs{value_outs_all, vsel};
  struct RowView {
    const std::vector<TapeRow>& all;
    const std::vector<uint32_t>& sel;
    size_t size() const { return sel.size(); }
    const TapeRow& operator[](size_t i) const { return all[sel[i]]; }
  } rows{rows_all, rsel};

  SetupLap lap;
  // ---- A. working copy of everything reachable from the roots ---------------
  std::vector<NodeId> roots;
  for (size_t v = 0; v < value_outs.size(); ++v) roots.push_back(value_outs[v].node);
  for (size_t r = 0; r < rows.size(); ++r) roots.push_back(rows[r].root);

  std::vector<NodeId> reach;
  {
    std::vector<uint8_t>& seen = scratch->seen;
    std::vector<NodeId> stack;
    for (NodeId r : roots)
      if (r != kNull && !seen[r]) {
        seen[r] = 1;
        stack.push_back(r);
      }
    while (!stack.empty()) {
      NodeId n = stack.back();
      stack.pop_back();
      reach.push_back(n);
      for (NodeId a : {g.a0[n], g.a1[n]})
        if (a != kNull && !seen[a]) {
          seen[a] = 1;
          stack.push_back(a);
        }
    }
    // ascending ids (children first): read off the marks instead of sorting the visit order — unless the set is a
    // small part of the graph (a family's representative)
    const size_t count = reach.size();
    if (count * 24 < seen.size()) {
      std::sort(reach.begin(), reach.end());
      for (NodeId n : reach) seen[n] = 0;
    } else {
      reach.clear();
      reach.reserve(count);
      for (size_t n = 0; n < seen.size(); ++n)
        if (seen[n]) {
          reach.push_back(static_cast<NodeId>(n));
          seen[n] = 0;
        }
    }
  }
  // graph node -> working-copy node (-1: not reachable); a flat table, the lookups are hot
  struct NodeTable {
    std::vector<int32_t>& v;
    int32_t at(NodeId n) const { return v[n]; }
    int32_t& operator[](NodeId n) { return v[n]; }
  } to_cg{scratch->to_cg};
  // (blank again when the call is over, whichever way it ends)
  struct Blank {
    std::vector<int32_t>& v;
    const std::vector<NodeId>& touched;
    ~Blank() {
      for (NodeId n : touched) v[n] = -1;
    }
  } blank_to_cg{scratch->to_cg, reach};
  CG cg;
  {
    // Common-subexpression elimination while copying (ascending node ids = children first):
    // an interior node with the same opcode and the same (already merged) operands as an
    // earlier one IS that node.  Operator overloading creates such duplicates wholesale —
    // every `cos(theta)` in a dynamics function is a node of its own, and gradient_tree
    // repeats whole subterms — and each costs a forward evaluation and an adjoint slot per
    // row that reaches it.  Values are unchanged bit for bit; an adjoint that used to be
    // propagated separately through the duplicates is now summed first.
    bool cse = opt.cse;
    if (const char* env = std::getenv("SLPX_TAPE_CSE")) cse = env[0] != '0';
    // open addressing, keys are never 0 (the opcode byte of an interior node is not): a node-based
    // map was a third of the whole tape compilation
    struct FlatMap {
      std::vector<uint64_t> keys;
      std::vector<int32_t> vals;
      uint64_t mask;
      explicit FlatMap(size_t n) {
        size_t cap = 16;
        while (cap < 2 * n) cap <<= 1;
        keys.assign(cap, 0);
        vals.assign(cap, 0);
        mask = cap - 1;
      }
      // the slot of `key` (fresh: it was not there and has been claimed)
      int32_t* find_or_claim(uint64_t key, bool& fresh) {
        uint64_t h = key * 0x9e3779b97f4a7c15ull;
        h ^= h >> 29;
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
          if (keys[i] == key) {
            fresh = false;
            return &vals[i];
          }
          if (keys[i] == 0) {
            keys[i] = key;
            fresh = true;
            return &vals[i];
          }
        }
      }
    } seen(cse ? reach.size() : 0);
    for (NodeId n : reach) {
      int32_t l = g.a0[n] == kNull ? -1 : to_cg.at(g.a0[n]);
      int32_t r = g.a1[n] == kNull ? -1 : to_cg.at(g.a1[n]);
      if (cse && l >= 0 && static_cast<uint32_t>(l) < (1u << 28) && static_cast<uint32_t>(r + 1) < (1u << 28)) {
        // a + b and a * b are exactly commutative in IEEE arithmetic: one key for both orders
        int32_t kl = l, kr = r;
        if ((g.op[n] == OP_ADD || g.op[n] == OP_MUL) && kr >= 0 && kr < kl) std::swap(kl, kr);
        const uint64_t key = (static_cast<uint64_t>(g.op[n]) << 56) | (static_cast<uint64_t>(kl) << 28) |
                             static_cast<uint64_t>(kr + 1);
        bool fresh = false;
        int32_t* slot = seen.find_or_claim(key | (1ull << 63), fresh);
        if (!fresh) {
          to_cg[n] = *slot;
          continue;
        }
        *slot = to_cg[n] = cg.add(g.op[n], l, r, n);
        continue;
      }
      to_cg[n] = cg.add(g.op[n], l, r, n);
    }
  }
  const std::vector<int32_t>& input_idx = scratch->input_idx;
  prog.n_inputs = std::max(prog.n_inputs, scratch->n_inputs);

  lap("  tape: A working copy");
  // ---- B/C. rebalance long left-deep ADD chains ---------------------------
  if (opt.rebalance_sums) {
    std::vector<int32_t> uses(cg.size(), 0);
    for (size_t n = 0; n < cg.size(); ++n) {
      if (cg.a0[n] >= 0) ++uses[cg.a0[n]];
      if (cg.a1[n] >= 0) ++uses[cg.a1[n]];
    }
    for (NodeId r : roots)
      if (r != kNull) uses[to_cg.at(r)] += 2;  // roots are never interior chain links
    const size_t n0 = cg.size();
    std::vector<uint8_t> consumed(n0, 0);
    for (int32_t top = static_cast<int32_t>(n0) - 1; top >= 0; --top) {
      if (cg.op[top] != OP_ADD || consumed[top]) continue;
      // walk down the left spine
      std::vector<int32_t> terms;  // collected right operands, top-down
      int32_t cur = top;
      while (true) {
        terms.push_back(cg.a1[cur]);
        int32_t next = cg.a0[cur];
        if (next >= 0 && cg.op[next] == OP_ADD && uses[next] == 1 && !consumed[next]) {
          consumed[next] = 1;
          cur = next;
        } else {
          terms.push_back(next);
          break;
        }
      }
      if (terms.size() < opt.rebalance_min_terms) continue;
      std::reverse(terms.begin(), terms.end());  // original left-to-right order
      // pairwise tree, keeping left-to-right term order
      std::vector<int32_t> level = terms;
      while (level.size() > 2) {
        std::vector<int32_t> next;
        for (size_t i = 0; i + 1 < level.size(); i += 2)
          next.push_back(cg.add(OP_ADD, level[i], level[i + 1], kNull));
        if (level.size() & 1) next.push_back(level.back());
        level.swap(next);
      }
      cg.a0[top] = level[0];
      cg.a1[top] = level[1];
    }
  }

  lap("  tape: B/C rebalance");
  // ---- liveness + levels (children may now have larger ids than parents) -----
  const size_t ncg = cg.size();
  std::vector<uint8_t> live(ncg, 0);
  std::vector<int32_t> order;  // children-before-parents order of live nodes
  {
    std::vector<std::pair<int32_t, int>> stack;  // (node, state)
    for (NodeId r : roots) {
      if (r == kNull) continue;
      int32_t c = to_cg.at(r);
      if (live[c]) continue;
      stack.push_back({c, 0});
      live[c] = 1;
      while (!stack.empty()) {
        auto& [n, st] = stack.back();
        int32_t child = st == 0 ? cg.a0[n] : (st == 1 ? cg.a1[n] : -2);
        if (child == -2) {
          order.push_back(n);
          stack.pop_back();
          continue;
        }
        ++st;
        if (child >= 0 && !live[child]) {
          live[child] = 1;
          stack.push_back({child, 0});
        }
      }
    }
  }
  std::vector<int32_t> level(ncg, 0);
  for (int32_t n : order) {
    if (cg.is_leaf(n)) continue;
    int32_t l = level[cg.a0[n]];
    if (cg.a1[n] >= 0) l = std::max(l, level[cg.a1[n]]);
    level[n] = l + 1;
  }

  lap("  tape: liveness + levels");
  // ---- rows: reference order, useful sets, slots, edges -------------------------
  struct Slot {
    int32_t row, node, level;
    uint32_t edge_begin, edge_end;  // into row-local edge storage
    int32_t out_dst;                // -1 if not an output
  };
  struct REdge {
    int32_t parent_slot;  // global slot id
    int32_t parent_node;
    int side;
  };
  std::vector<Slot> slots;
  // edges INTO each slot, CSR over global slot ids (a slot's edges keep the order in which
  // the reference's sweep meets the parents: list order, left operand before right)
  std::vector<uint32_t> slot_edge_ptr{0};
  std::vector<REdge> slot_edge_data;
  auto n_slot_edges = [&](int32_t sl) { return static_cast<size_t>(slot_edge_ptr[sl + 1] - slot_edge_ptr[sl]); };
  std::vector<uint8_t> need_dl(ncg, 0), need_dr(ncg, 0);
  std::vector<int32_t> row_root_slot(rows.size(), -1);
  {
    std::vector<int32_t> slot_of(ncg, -1);
    std::vector<int32_t> out_dst(ncg, -1);
    std::vector<uint8_t> useful(ncg, 0);
    std::vector<uint32_t> fill_pos;
    for (size_t ri = 0; ri < rows.size(); ++ri) {
      const TapeRow& row = rows[ri];
      if (row.root == kNull) continue;
      int32_t root = to_cg.at(row.root);
      for (auto& o : row.outputs) {
        const int32_t w = to_cg.at(o.wrt);
        if (w >= 0) out_dst[w] = o.dst;
      }
      std::vector<int32_t> top = reference_order(cg, root);
      for (auto it = top.rbegin(); it != top.rend(); ++it) {
        int32_t n = *it;
        bool u = out_dst[n] >= 0;
        if (cg.a0[n] >= 0 && useful[cg.a0[n]]) u = true;
        if (cg.a1[n] >= 0 && useful[cg.a1[n]]) u = true;
        useful[n] = u;
      }
      const size_t row_slot0 = slots.size();
      for (int32_t n : top) {
        if (!useful[n]) continue;
        slot_of[n] = static_cast<int32_t>(slots.size());
        slots.push_back({static_cast<int32_t>(ri), n, 0, 0, 0, out_dst[n]});
      }
      row_root_slot[ri] = slot_of[root];
      // count, then fill (two passes over the row's list keep the per-slot edge order)
      fill_pos.assign(slots.size() - row_slot0 + 1, 0);
      for (int32_t p : top) {
        if (!useful[p] || cg.is_leaf(p)) continue;
        const int32_t l = cg.a0[p], r = cg.a1[p];
        if (l >= 0 && useful[l]) ++fill_pos[slot_of[l] - row_slot0 + 1];
        if (r >= 0 && useful[r]) ++fill_pos[slot_of[r] - row_slot0 + 1];
      }
      const uint32_t edge0 = static_cast<uint32_t>(slot_edge_data.size());
      for (size_t k = 1; k < fill_pos.size(); ++k) {
        fill_pos[k] += fill_pos[k - 1];
        slot_edge_ptr.push_back(edge0 + fill_pos[k]);
      }
      slot_edge_data.resize(edge0 + fill_pos.back());
      for (int32_t p : top) {
        if (!useful[p] || cg.is_leaf(p)) continue;
        int32_t ps = slot_of[p];
        int32_t l = cg.a0[p], r = cg.a1[p];
        if (l >= 0 && useful[l]) {
          slot_edge_data[edge0 + fill_pos[slot_of[l] - row_slot0]++] = {ps, p, 0};
          need_dl[p] = 1;
          slots[slot_of[l]].level = std::max(slots[slot_of[l]].level, slots[ps].level + 1);
        }
        if (r >= 0 && useful[r]) {
          slot_edge_data[edge0 + fill_pos[slot_of[r] - row_slot0]++] = {ps, p, 1};
          need_dr[p] = 1;
          slots[slot_of[r]].level = std::max(slots[slot_of[r]].level, slots[ps].level + 1);
        }
      }
      for (int32_t n : top) {
        useful[n] = 0;
        slot_of[n] = -1;
        out_dst[n] = -1;
      }
      for (auto& o : row.outputs) {
        const int32_t w = to_cg.at(o.wrt);
        if (w >= 0) out_dst[w] = -1;
      }
    }
  }
  // NOTE on slot levels: parents precede children in `top`, and every edge into a
  // child is visited after all edges into its parent were visited, because the
  // parent's own level is final once it has been popped in reference order (all
  // its incoming edges come from nodes earlier in the list).

  lap("  tape: rows: slots + edges");
  // ---- components of interior nodes ------------------------------------------
  // A node whose operands are all leaves (e.g. the `-y_j` of the Lagrangian, which
  // both stage k and stage k+1 reference) is REPLICATED into every component that
  // uses it instead of gluing those components together: recomputing one cheap op
  // keeps the stages independent.
  std::vector<uint8_t> is_root(ncg, 0), replicable(ncg, 0);
  for (NodeId r : roots)
    if (r != kNull) is_root[to_cg.at(r)] = 1;
  // ... and so is a node on top of such a node and leaves (depth 2): the adjoint seed of a
  // SCALED constraint row, (-y_j) * d_j, is shared by the row's own stage and — through the
  // gradient of the next stage's state — by its neighbour; without this the feasibility
  // restoration model (rows d_j c_j(x) - p_j + n_j) collapses into ONE component of 214 k
  // nodes that only fits HBM scratch (0.68 ms per sweep at cart-pole N=500).
  {
    std::vector<uint8_t> depth(ncg, 0);  // 0: not replicable
    for (int32_t n : order) {  // children first
      if (cg.is_leaf(n) || is_root[n]) continue;
      uint8_t d = 1;
      bool ok = true;
      int interior_operands = 0;
      for (int32_t a : {cg.a0[n], cg.a1[n]}) {
        if (a < 0 || cg.is_leaf(a)) continue;
        ++interior_operands;
        if (depth[a] == 0) ok = false;
        else d = std::max<uint8_t>(d, static_cast<uint8_t>(depth[a] + 1));
      }
      // (one interior operand only: a node joining TWO replicable nodes is the first level
      // of a sum tree — replicating those dissolves the small identical components, e.g. the
      // term groups of a separable cost, that the template kernels live on)
      if (ok && d <= 2 && interior_operands <= 1) {
        depth[n] = d;
        replicable[n] = 1;
      }
    }
  }
  UnionFind uf(ncg);
  for (int32_t n : order) {
    if (cg.is_leaf(n) || replicable[n]) continue;
    for (int32_t a : {cg.a0[n], cg.a1[n]})
      if (a >= 0 && !cg.is_leaf(a) && !replicable[a]) uf.unite(n, a);
  }
  // component id -> list of interior nodes; stable order by smallest source id
  std::vector<int32_t> comp_of(ncg, -1);
  std::vector<std::vector<int32_t>> comp_nodes;
  {
    std::unordered_map<int32_t, int32_t> idx;
    std::vector<int32_t> interior;
    for (int32_t n : order)
      if (!cg.is_leaf(n) && !replicable[n]) interior.push_back(n);
    std::sort(interior.begin(), interior.end());
    for (int32_t n : interior) {
      int32_t r = uf.find(n);
      auto it = idx.find(r);
      if (it == idx.end()) {
        it = idx.emplace(r, static_cast<int32_t>(comp_nodes.size())).first;
        comp_nodes.emplace_back();
      }
      comp_of[n] = it->second;
      comp_nodes[it->second].push_back(n);
    }
  }
  const size_t ncomp = comp_nodes.size();
  std::vector<std::vector<int32_t>> comp_slots(ncomp);
  std::vector<std::vector<size_t>> comp_vouts(ncomp);
  std::vector<size_t> leaf_vouts;  // value outputs that are leaves
  for (size_t s = 0; s < slots.size(); ++s) {
    int32_t n = slots[s].node;
    int32_t c;
    if (!cg.is_leaf(n) && !replicable[n]) {
      c = comp_of[n];
    } else {
      // a leaf (or replicated) slot belongs to the component of its row's root
      // (the root is interior whenever the row has any edge)
      c = comp_of[to_cg.at(rows[slots[s].row].root)];
      if (c < 0) continue;  // row whose root is a leaf: no tape work
    }
    comp_slots[c].push_back(static_cast<int32_t>(s));
  }
  for (size_t v = 0; v < value_outs.size(); ++v) {
    int32_t n = to_cg.at(value_outs[v].node);
    if (cg.is_leaf(n)) leaf_vouts.push_back(v);
    else comp_vouts[comp_of[n]].push_back(v);
  }

  // LDS bytes of a component: working set (values, partials, adjoint slots) plus
  // the staged, 16-bit-packed program (8 B node records, 4 B edges, 2 B edge ptrs)
  auto comp_cost = [&](size_t c, size_t& n_leaf_est) {
    // leaves are counted per component (upper bound when packing several)
    size_t leaves = 0;
    for (int32_t n : comp_nodes[c]) {
      if (cg.a0[n] >= 0 && cg.is_leaf(cg.a0[n])) ++leaves;
      if (cg.a1[n] >= 0 && cg.is_leaf(cg.a1[n])) ++leaves;
    }
    n_leaf_est = leaves;
    size_t edges = 0;
    for (int32_t sl : comp_slots[c]) edges += n_slot_edges(sl);
    return 8 * leaves + (24 + 8) * comp_nodes[c].size() + (8 + 2) * comp_slots[c].size() +
           4 * edges + 64;
  };

  lap("  tape: components");
  // ---- pack components into tasks -----------------------------------------------
  const size_t small_cap = opt.small_lds_bytes, large_cap = opt.large_lds_bytes;
  struct Pack {
    std::vector<size_t> comps;
    size_t cost = 0;
    int cls = 0;  // 0 small, 1 large, 2 global
    size_t leaf_begin = 0, leaf_end = 0;  // leaf-only task: its slice of leaf_vouts
  };
  std::vector<Pack> packs;
  {
    // Components that come in large families of the same shape (the stages of a
    // transcribed OCP, the term groups of a separable cost) get a task EACH: identical
    // tasks share one program copy and run as instances of one generated template kernel
    // (tape_jit.hpp); bin-packing them together would destroy exactly that regularity.
    std::unordered_map<uint64_t, uint32_t> family;
    std::vector<uint64_t> signature(ncomp);
    for (size_t c = 0; c < ncomp; ++c) {
      uint64_t h = 1469598103934665603ull;
      auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
      mix(comp_nodes[c].size());
      mix(comp_slots[c].size());
      mix(comp_vouts[c].size());
      size_t edges = 0;
      for (int32_t sl : comp_slots[c]) edges += n_slot_edges(sl);
      mix(edges);
      uint64_t ops = 0;
      for (int32_t nd : comp_nodes[c]) ops += static_cast<uint64_t>(cg.op[nd]) * 131u + static_cast<uint64_t>(level[nd]);
      mix(ops);
      signature[c] = h;
      ++family[h];
    }
    // the estimate of comp_cost ignores padding and shared-leaf effects: keep a margin so a
    // packed task does not spill into the (slower, 256-thread) large class by accident
    const size_t pack_cap = small_cap - small_cap / 8;
    Pack cur;
    for (size_t c = 0; c < ncomp; ++c) {
      size_t le;
      size_t cost = comp_cost(c, le);
      if (cost > small_cap) {
        Pack big;
        big.comps.push_back(c);
        big.cost = cost;
        big.cls = cost > large_cap ? 2 : 1;
        packs.push_back(std::move(big));
        continue;
      }
      // (8: the term groups of a separable cost over 300-500 steps are 9-15 of a kind; packed into
      // one interpreted task they took 11-20 us against 7 us for the whole generated kernel)
      if (family[signature[c]] >= kTapeFamilyMin && comp_nodes[c].size() >= 16) {
        Pack own;
        own.comps.push_back(c);
        own.cost = cost;
        packs.push_back(std::move(own));
        continue;
      }
      if (cur.cost + cost > pack_cap && !cur.comps.empty()) {
        packs.push_back(std::move(cur));
        cur = Pack{};
      }
      cur.comps.push_back(c);
      cur.cost += cost;
    }
    if (!cur.comps.empty()) packs.push_back(std::move(cur));
  }
  // Value outputs that are bare leaves (a bound like x >= 0 prunes to the variable itself)
  // need no program at all: copy tasks, cut so that each stays a small 64-thread task
  // (12 B of LDS-staged state per leaf) instead of one task that outgrows the small class
  // with the horizon (N=5000: one 256-thread launch of 26 us behind everything else).
  {
    const size_t chunk = std::max<size_t>(64, small_cap / 16);
    for (size_t lb = 0; lb < leaf_vouts.size(); lb += chunk) {
      Pack lp;
      lp.cls = 0;
      lp.leaf_begin = lb;
      lp.leaf_end = std::min(leaf_vouts.size(), lb + chunk);
      packs.push_back(std::move(lp));  // leaf-only task, filled below
    }
  }

  lap("  tape: pack");
  // ---- emit tasks -----------------------------------------------------------------
  std::unordered_map<double, uint32_t> const_pool;
  std::unordered_map<NodeId, uint32_t> param_slot;
  auto const_index = [&](double v) {
    auto it = const_pool.find(v);
    if (it != const_pool.end()) return it->second;
    uint32_t i = static_cast<uint32_t>(prog.consts.size());
    prog.consts.push_back(v);
    const_pool.emplace(v, i);
    return i;
  };
  auto leaf_binding = [&](int32_t n) -> uint32_t {
    NodeId s = cg.src[n];
    if (cg.op[n] == OP_CONST) return kLeafConstFlag | const_index(g.val[s]);
    const int32_t in_idx = input_idx[s];
    // A free Variable that is not a decision variable acts as a parameter: in the
    // reference it is just a leaf holding its current value during solve().  It gets a
    // constant slot of its own, initialised with its value at compile time and
    // refreshable afterwards (TapeProgram::params).
    if (in_idx < 0) {
      auto pit = param_slot.find(s);
      if (pit == param_slot.end()) {
        const uint32_t i = static_cast<uint32_t>(prog.consts.size());
        prog.consts.push_back(g.val[s]);
        prog.params.emplace_back(s, i);
        pit = param_slot.emplace(s, i).first;
      }
      return kLeafConstFlag | pit->second;
    }
    return static_cast<uint32_t>(in_idx);
  };

  // Parameters get their slots FIRST and in node order (`param_order`: the caller's list, which may name more than
  // this program reaches), not in the order in which tasks happen to meet them: the slot of "parameter j of stage i"
  // is then base + stride * i whatever the packing, and the generated kernel computes such words instead of loading
  // them — with the same generic source for every horizon (tape_jit.cpp).
  {
    auto claim = [&](NodeId s) {
      if (param_slot.count(s)) return;
      const uint32_t i = static_cast<uint32_t>(prog.consts.size());
      prog.consts.push_back(g.val[s]);
      prog.params.emplace_back(s, i);
      param_slot.emplace(s, i);
    };
    if (param_order != nullptr) {
      for (NodeId s : *param_order) claim(s);
    } else {
      for (size_t n = 0; n < ncg; ++n)
        if (live[n] && cg.is_leaf(static_cast<int32_t>(n)) && cg.op[n] == OP_VAR && input_idx[cg.src[n]] < 0) claim(cg.src[n]);
    }
  }
  std::unordered_multimap<uint64_t, uint32_t> templates;  // structure hash -> first task with it
  std::vector<int32_t> local_of(ncg, -1);       // CG node -> local value index
  std::vector<int32_t> local_slot(slots.size(), -1);
  for (size_t pi = 0; pi < packs.size(); ++pi) {
    Pack& pk = packs[pi];
    // Every task's slice of every program array starts on a 16-byte boundary, so the
    // kernel can stage it into LDS with unrolled 16-byte loads.
    auto pad = [](auto& v, size_t multiple) {
      while (v.size() % multiple) v.push_back({});
    };
    pad(prog.leaf_src, 4);
    TapeTask t{};
    t.leaf_off = static_cast<uint32_t>(prog.leaf_src.size());
    // the structural part of the task's program is built locally first: tasks with
    // byte-identical structure (the stages of a transcribed OCP) share ONE copy of it
    std::vector<uint32_t> L_rec, L_lvl, L_eptr, L_slvl;
    std::vector<uint16_t> L_rec16, L_eptr16, L_edges16;
    std::vector<TapeEdge> L_edges;
    t.vout_off = static_cast<uint32_t>(prog.vout_src.size());
    t.jout_off = static_cast<uint32_t>(prog.jout_slot.size());

    std::vector<int32_t> nodes, tslots, leaves;
    std::vector<size_t> vouts;
    const bool leaf_task = pk.comps.empty();
    if (leaf_task) {
      vouts.assign(leaf_vouts.begin() + pk.leaf_begin, leaf_vouts.begin() + pk.leaf_end);
      for (size_t v : vouts) leaves.push_back(to_cg.at(value_outs[v].node));
    }
    for (size_t c : pk.comps) {
      nodes.insert(nodes.end(), comp_nodes[c].begin(), comp_nodes[c].end());
      tslots.insert(tslots.end(), comp_slots[c].begin(), comp_slots[c].end());
      vouts.insert(vouts.end(), comp_vouts[c].begin(), comp_vouts[c].end());
    }
    {  // private copies of the replicated nodes this task consumes (and of what THEY consume)
      std::vector<int32_t> extra, frontier = nodes;
      while (!frontier.empty()) {
        std::vector<int32_t> next;
        for (int32_t n : frontier)
          for (int32_t a : {cg.a0[n], cg.a1[n]})
            if (a >= 0 && !cg.is_leaf(a) && replicable[a]) next.push_back(a);
        std::sort(next.begin(), next.end());
        next.erase(std::unique(next.begin(), next.end()), next.end());
        frontier.clear();
        for (int32_t a : next)
          if (!std::binary_search(extra.begin(), extra.end(), a)) frontier.push_back(a);
        extra.insert(extra.end(), frontier.begin(), frontier.end());
        std::sort(extra.begin(), extra.end());
      }
      nodes.insert(nodes.end(), extra.begin(), extra.end());
    }
    for (int32_t n : nodes)
      for (int32_t a : {cg.a0[n], cg.a1[n]})
        if (a >= 0 && cg.is_leaf(a)) leaves.push_back(a);
    std::sort(leaves.begin(), leaves.end());
    leaves.erase(std::unique(leaves.begin(), leaves.end()), leaves.end());
    std::stable_sort(nodes.begin(), nodes.end(),
                     [&](int32_t a, int32_t b) { return level[a] < level[b]; });
    std::stable_sort(tslots.begin(), tslots.end(), [&](int32_t a, int32_t b) {
      return slots[a].level < slots[b].level;
    });

    t.n_leaf = static_cast<uint32_t>(leaves.size());
    t.n_node = static_cast<uint32_t>(nodes.size());
    t.n_slot = static_cast<uint32_t>(tslots.size());
    for (size_t i = 0; i < leaves.size(); ++i) {
      local_of[leaves[i]] = static_cast<int32_t>(i);
      prog.leaf_src.push_back(leaf_binding(leaves[i]));
    }
    for (size_t i = 0; i < nodes.size(); ++i) local_of[nodes[i]] = static_cast<int32_t>(t.n_leaf + i);
    // node records + forward levels
    {
      int32_t cur_level = -1;
      for (size_t i = 0; i < nodes.size(); ++i) {
        int32_t n = nodes[i];
        if (level[n] != cur_level) {  // one group per DISTINCT level present in the task
          L_lvl.push_back(static_cast<uint32_t>(i));
          cur_level = level[n];
        }
        if (!op_is_basic(static_cast<Opcode>(cg.op[n]))) prog.basic_ops = false;
        uint32_t rec = cg.op[n] | (need_dl[n] ? 0x100u : 0u) | (need_dr[n] ? 0x200u : 0u);
        L_rec.push_back(rec);
        L_rec.push_back(static_cast<uint32_t>(local_of[cg.a0[n]]));
        L_rec.push_back(cg.a1[n] >= 0 ? static_cast<uint32_t>(local_of[cg.a1[n]])
                                      : static_cast<uint32_t>(local_of[cg.a0[n]]));
        // 16-bit packed copy for the LDS-staged kernel (unused by GLOBAL tasks)
        const size_t r3 = L_rec.size() - 3;
        L_rec16.push_back(static_cast<uint16_t>(rec));
        L_rec16.push_back(static_cast<uint16_t>(L_rec[r3 + 1]));
        L_rec16.push_back(static_cast<uint16_t>(L_rec[r3 + 2]));
        L_rec16.push_back(0);
      }
      L_lvl.push_back(static_cast<uint32_t>(nodes.size()));
      t.n_lvl = static_cast<uint32_t>(L_lvl.size() - 1);
      prog.max_levels = std::max(prog.max_levels, t.n_lvl);
    }
    // slots + reverse levels + edges
    {
      for (size_t i = 0; i < tslots.size(); ++i) local_slot[tslots[i]] = static_cast<int32_t>(i);
      int32_t cur_level = -1;
      uint32_t edge_count = 0;
      for (size_t i = 0; i < tslots.size(); ++i) {
        const Slot& s = slots[tslots[i]];
        if (s.level != cur_level) {
          L_slvl.push_back(static_cast<uint32_t>(i));
          cur_level = s.level;
        }
        L_eptr.push_back(edge_count);
        L_eptr16.push_back(static_cast<uint16_t>(edge_count));
        for (uint32_t ek = slot_edge_ptr[tslots[i]]; ek < slot_edge_ptr[tslots[i] + 1]; ++ek) {
          const REdge& e = slot_edge_data[ek];
          uint32_t pn = static_cast<uint32_t>(local_of[e.parent_node]) - t.n_leaf;
          L_edges.push_back({static_cast<uint32_t>(local_slot[e.parent_slot]),
                             2u * pn + static_cast<uint32_t>(e.side)});
          L_edges16.push_back(static_cast<uint16_t>(local_slot[e.parent_slot]));
          L_edges16.push_back(static_cast<uint16_t>(2u * pn + static_cast<uint32_t>(e.side)));
          ++edge_count;
        }
        if (s.out_dst >= 0) {
          prog.jout_slot.push_back(static_cast<uint32_t>(i));
          prog.jout_dst.push_back(static_cast<uint32_t>(s.out_dst));
          prog.jout_scale.push_back(rows[s.row].scale_idx);
        }
      }
      L_eptr.push_back(edge_count);
      L_eptr16.push_back(static_cast<uint16_t>(edge_count));
      t.n_edge = edge_count;
      L_slvl.push_back(static_cast<uint32_t>(tslots.size()));
      t.n_slvl = static_cast<uint32_t>(L_slvl.size() - 1);
      prog.max_slot_levels = std::max(prog.max_slot_levels, t.n_slvl);
      prog.total_edges += edge_count;
    }
    {  // share the structure with an identical earlier task, else append it
      uint64_t h = 1469598103934665603ull;
      auto mix = [&](const void* data, size_t bytes) {  // all arrays are whole 32-bit words
        const uint32_t* w = static_cast<const uint32_t*>(data);
        for (size_t k = 0; k < bytes / 4; ++k) {
          h = (h ^ w[k]) * 1099511628211ull;
          h ^= h >> 29;
        }
      };
      const uint32_t dims[3] = {t.n_leaf, t.n_node, t.n_slot};
      mix(dims, sizeof(dims));
      mix(L_rec.data(), L_rec.size() * 4);
      mix(L_lvl.data(), L_lvl.size() * 4);
      mix(L_eptr.data(), L_eptr.size() * 4);
      mix(L_slvl.data(), L_slvl.size() * 4);
      mix(L_edges.data(), L_edges.size() * sizeof(TapeEdge));
      auto same = [&](const TapeTask& o) {
        if (o.n_leaf != t.n_leaf || o.n_node != t.n_node || o.n_slot != t.n_slot || o.n_lvl != t.n_lvl ||
            o.n_slvl != t.n_slvl || o.n_edge != t.n_edge)
          return false;
        return std::equal(L_rec.begin(), L_rec.end(), prog.node_rec.begin() + 3 * size_t(o.node_off)) &&
               std::equal(L_lvl.begin(), L_lvl.end(), prog.lvl_ptr.begin() + o.lvl_off) &&
               std::equal(L_eptr.begin(), L_eptr.end(), prog.slot_edge_ptr.begin() + o.slot_off) &&
               std::equal(L_slvl.begin(), L_slvl.end(), prog.slvl_ptr.begin() + o.slvl_off) &&
               std::equal(L_edges.begin(), L_edges.end(), prog.edges.begin() + o.edge_off,
                          [](const TapeEdge& a, const TapeEdge& b) {
                            return a.parent_slot == b.parent_slot && a.partial == b.partial;
                          });
      };
      int64_t shared = -1;
      auto range = templates.equal_range(h);
      for (auto it = range.first; it != range.second && shared < 0; ++it)
        if (same(prog.tasks[it->second])) shared = it->second;
      if (shared >= 0) {
        const TapeTask& o = prog.tasks[shared];
        t.node_off = o.node_off;
        t.lvl_off = o.lvl_off;
        t.slot_off = o.slot_off;
        t.slvl_off = o.slvl_off;
        t.edge_off = o.edge_off;
        ++prog.shared_tasks;
      } else {
        // every slice starts on a 16-byte boundary (unrolled 16-byte staging loads)
        while ((prog.node_rec.size() / 3) % 2) {  // 8-byte packed records: even node offset
          for (int k = 0; k < 3; ++k) prog.node_rec.push_back(0);
          for (int k = 0; k < 4; ++k) prog.node_rec16.push_back(0);
        }
        while (prog.edges.size() % 4) {  // 4-byte packed edges
          prog.edges.push_back({0, 0});
          prog.edges16.push_back(0);
          prog.edges16.push_back(0);
        }
        while (prog.slot_edge_ptr.size() % 8) {  // 2-byte packed edge pointers
          prog.slot_edge_ptr.push_back(0);
          prog.slot_edge_ptr16.push_back(0);
        }
        pad(prog.lvl_ptr, 4);
        pad(prog.slvl_ptr, 4);
        t.node_off = static_cast<uint32_t>(prog.node_rec.size() / 3);
        t.lvl_off = static_cast<uint32_t>(prog.lvl_ptr.size());
        t.slot_off = static_cast<uint32_t>(prog.slot_edge_ptr.size());
        t.slvl_off = static_cast<uint32_t>(prog.slvl_ptr.size());
        t.edge_off = static_cast<uint32_t>(prog.edges.size());
        prog.node_rec.insert(prog.node_rec.end(), L_rec.begin(), L_rec.end());
        prog.node_rec16.insert(prog.node_rec16.end(), L_rec16.begin(), L_rec16.end());
        prog.lvl_ptr.insert(prog.lvl_ptr.end(), L_lvl.begin(), L_lvl.end());
        prog.slot_edge_ptr.insert(prog.slot_edge_ptr.end(), L_eptr.begin(), L_eptr.end());
        prog.slot_edge_ptr16.insert(prog.slot_edge_ptr16.end(), L_eptr16.begin(), L_eptr16.end());
        prog.slvl_ptr.insert(prog.slvl_ptr.end(), L_slvl.begin(), L_slvl.end());
        prog.edges.insert(prog.edges.end(), L_edges.begin(), L_edges.end());
        prog.edges16.insert(prog.edges16.end(), L_edges16.begin(), L_edges16.end());
        templates.emplace(h, static_cast<uint32_t>(prog.tasks.size()));
      }
    }
    for (size_t v : vouts) {
      prog.vout_src.push_back(static_cast<uint32_t>(local_of[to_cg.at(value_outs[v].node)]));
      prog.vout_dst.push_back(static_cast<uint32_t>(value_outs[v].dst));
      prog.vout_scale.push_back(value_outs[v].scale_idx);
    }
    t.n_vout = static_cast<uint32_t>(prog.vout_src.size() - t.vout_off);
    t.n_jout = static_cast<uint32_t>(prog.jout_slot.size() - t.jout_off);
    if (trace != nullptr) {
      TapeTrace::Task tt;
      for (int32_t lf : leaves) tt.leaf_nodes.push_back(cg.src[lf]);
      for (size_t v : vouts) tt.vouts.push_back(vsel[v]);
      for (int32_t sl : tslots)
        if (slots[sl].out_dst >= 0) tt.jouts.emplace_back(rsel[slots[sl].row], cg.src[slots[sl].node]);
      for (size_t c : pk.comps) tt.comp_nodes += static_cast<uint32_t>(comp_nodes[c].size());
      tt.n_comps = static_cast<uint32_t>(pk.comps.size());
      trace->tasks.push_back(std::move(tt));
    }
    t.lds_doubles = t.n_leaf + 3 * t.n_node + t.n_slot;
    auto q16 = [](uint32_t count, uint32_t per16) { return 16u * ((count + per16 - 1) / per16); };
    t.lds_bytes = q16(t.n_node, 2) + q16(t.n_edge, 4) + q16(t.n_slot + 1, 8) + q16(t.n_lvl + 1, 4) +
                  q16(t.n_slvl + 1, 4) + q16(t.n_leaf, 4) + 8 * t.lds_doubles;
    int cls = pk.cls;
    if (cls == 0 && t.lds_bytes > small_cap) cls = 1;
    if (cls == 1 && t.lds_bytes > large_cap) cls = 2;
    // 16-bit packing limits of the staged kernel
    if (t.n_leaf + t.n_node > 65535u || t.n_slot > 65535u || t.n_edge > 65535u ||
        2u * t.n_node > 65535u)
      cls = 2;
    uint32_t ti = static_cast<uint32_t>(prog.tasks.size());
    if (cls == 0) {
      prog.small_tasks.push_back(ti);
      prog.small_lds_bytes = std::max(prog.small_lds_bytes, t.lds_bytes);
    } else if (cls == 1) {
      prog.large_tasks.push_back(ti);
      prog.large_lds_bytes = std::max(prog.large_lds_bytes, t.lds_bytes);
    } else {
      prog.global_tasks.push_back(ti);
      t.scratch_off = static_cast<uint32_t>(prog.global_scratch_doubles);
      prog.global_scratch_doubles += t.lds_doubles;
    }
    if (trace != nullptr) trace->tasks.back().cls = cls;
    prog.tasks.push_back(t);
    prog.total_nodes += t.n_node;
    prog.total_slots += t.n_slot;
    prog.total_leaves += t.n_leaf;
  }
  if (tail_padding) add_tail_padding(prog);
  lap("  tape: emit");
  return prog;
}

// ---------------------------------------------------------------------------
// Families ("compile one stage, replicate").  A transcribed optimal-control problem is the same
// stage N times over: the same expressions on shifted variables.  The flat compiler above used to put every one of
// them through its passes — working copy, adjoint slots and edges, emission, 0.4 s at cart-pole N=1000 — only to
// find at the very end that the tasks are byte-identical.  Here the stages are found FIRST, on the raw graph, by a
// few linear passes:
//   1. components of interior nodes (the flat compiler's rule: nodes joined through interior operands; a node on
//      top of leaves only — depth <= 2, one interior operand — is private to each of its users);
//   2. every component's reachable set in ascending node order — its members, the private nodes it uses, its
//      leaves — written down as a sequence (opcode, position of operand 0, position of operand 1) followed by its
//      rows (position of the root, positions of the wrt leaves) and value outputs.  The flat compiler is a
//      deterministic function of exactly that sequence (node numbers enter only through their order), so two
//      components with EQUAL sequences compile to the same task, leaf for leaf, slot for slot;
//   3. one member of every family of kTapeFamilyMin or more goes through the flat compiler alone (with a trace of
//      which graph node every leaf, which row every output came from); the others are instantiated from it by
//      position: their leaf bindings and output destinations, nothing else;
//   4. what is in no family (unique components, rows of bare leaves, small packs) goes through the flat compiler
//      together, as before.
// The resulting program computes what the flat compiler's does, task for task (tests/test_tape_families_cpu.py:
// the same V to the bit); the order of the tasks in it differs.
// ---------------------------------------------------------------------------
// per-graph-node tables of the family passes, kept by the thread between compilations (a fresh 20 MB of them was
// mostly page faults)
struct FamilyNodeTables {
  std::vector<uint8_t> flag, depth;
  std::vector<int32_t> parent, comp_of, input_idx;
  std::vector<uint32_t> local;
  FlatScratch flat;
};
static FamilyNodeTables& family_node_tables() {
  static thread_local FamilyNodeTables tables;
  return tables;
}

bool tape_families_analyze(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                           const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows, TapeFamilySet& S) {
  SetupLap lap;
  const size_t G = g.size();
  constexpr uint8_t kReach = 1, kRoot = 2, kRepl = 4;
  // (tables of one entry per graph node, kept by the thread between compilations: a fresh 20 MB of them was
  // mostly page faults)
  FamilyNodeTables& tables = family_node_tables();
  std::vector<uint8_t>& flag = tables.flag;
  flag.assign(G, 0);
  // ---- 1. reachable nodes, roots ----
  // (operands have smaller numbers than the nodes that use them: ONE descending pass over the graph marks
  // everything below the roots — no stack, no second visit; a depth-first walk took twice as long at N=5000)
  {
    NodeId highest = -1;
    auto seed = [&](NodeId r) {
      if (r == kNull) return;
      flag[r] |= kRoot | kReach;
      highest = std::max(highest, r);
    };
    for (auto& v : value_outs) seed(v.node);
    for (auto& r : rows) seed(r.root);
    for (NodeId n = highest; n >= 0; --n) {
      if (!(flag[n] & kReach)) continue;
      const NodeId a = g.a0[n], b = g.a1[n];
      if (a != kNull) flag[a] |= kReach;
      if (b != kNull) flag[b] |= kReach;
    }
  }
  lap("  tape families:   reach");
  auto is_leaf = [&](NodeId n) { return g.a0[n] == kNull; };
  // ---- components (children have smaller numbers than their parents: ascending order is children first) ----
  std::vector<int32_t>& parent = tables.parent;  // union-find over interior, not private nodes
  if (parent.size() < G) parent.resize(G);
  auto find = [&](int32_t x) {
    while (parent[x] != x) {
      parent[x] = parent[parent[x]];
      x = parent[x];
    }
    return x;
  };
  {
    std::vector<uint8_t>& depth = tables.depth;
    depth.assign(G, 0);
    for (size_t n = 0; n < G; ++n) {
      parent[n] = static_cast<int32_t>(n);
      if (!(flag[n] & kReach) || is_leaf(static_cast<NodeId>(n))) continue;
      if (!(flag[n] & kRoot)) {
        uint8_t d = 1;
        bool ok = true;
        int interior_operands = 0;
        for (NodeId a : {g.a0[n], g.a1[n]}) {
          if (a == kNull || is_leaf(a)) continue;
          ++interior_operands;
          if (depth[a] == 0) ok = false;
          else d = std::max<uint8_t>(d, static_cast<uint8_t>(depth[a] + 1));
        }
        if (ok && d <= 2 && interior_operands <= 1) {
          depth[n] = d;
          flag[n] |= kRepl;
          continue;
        }
      }
      for (NodeId a : {g.a0[n], g.a1[n]})
        if (a != kNull && !is_leaf(a) && !(flag[a] & kRepl)) {
          const int32_t ra = find(a), rn = find(static_cast<int32_t>(n));
          if (ra != rn) parent[std::max(ra, rn)] = std::min(ra, rn);
        }
    }
  }
  lap("  tape families:   union");
  // component numbers in order of their smallest node; members by component, ascending
  std::vector<int32_t>& comp_of = tables.comp_of;
  comp_of.assign(G, -1);
  std::vector<uint32_t>& comp_start = S.comp_start;
  std::vector<NodeId>& members = S.members;
  comp_start.assign(1, 0);
  members.clear();
  {
    std::vector<uint32_t> count;
    for (size_t n = 0; n < G; ++n) {
      if (!(flag[n] & kReach) || (flag[n] & kRepl) || is_leaf(static_cast<NodeId>(n))) continue;
      const int32_t r = find(static_cast<int32_t>(n));
      if (comp_of[r] < 0) {  // (the representative is the smallest node of the component: met first)
        comp_of[r] = static_cast<int32_t>(count.size());
        count.push_back(0);
      }
      comp_of[n] = comp_of[r];
      ++count[comp_of[n]];
    }
    comp_start.resize(count.size() + 1);
    for (size_t c = 0; c < count.size(); ++c) comp_start[c + 1] = comp_start[c] + count[c];
    members.resize(comp_start.back());
    std::vector<uint32_t> fill(comp_start.begin(), comp_start.end() - 1);
    for (size_t n = 0; n < G; ++n)
      if (comp_of[n] >= 0) members[fill[comp_of[n]]++] = static_cast<NodeId>(n);
  }
  lap("  tape families:   numbering");
  const size_t ncomp = comp_start.size() - 1;
  S.ncomp = ncomp;
  S.graph_size = G;
  if (ncomp < kTapeFamilyMin) return false;
  // rows and value outputs by component (in the order of the lists); what belongs to none: a bare leaf
  std::vector<uint32_t>&crow_start = S.crow_start, &cvout_start = S.cvout_start, &crow = S.crow, &cvout = S.cvout,
                       &loose_rows = S.loose_rows, &loose_vouts = S.loose_vouts;
  crow_start.assign(ncomp + 1, 0);
  cvout_start.assign(ncomp + 1, 0);
  crow.clear(), cvout.clear(), loose_rows.clear(), loose_vouts.clear();
  {
    for (size_t ri = 0; ri < rows.size(); ++ri) {
      const NodeId r = rows[ri].root;
      if (r != kNull && comp_of[r] >= 0) ++crow_start[comp_of[r] + 1];
      else loose_rows.push_back(static_cast<uint32_t>(ri));
    }
    for (size_t v = 0; v < value_outs.size(); ++v) {
      const NodeId r = value_outs[v].node;
      if (r != kNull && comp_of[r] >= 0) ++cvout_start[comp_of[r] + 1];
      else loose_vouts.push_back(static_cast<uint32_t>(v));
    }
    for (size_t c = 0; c < ncomp; ++c) {
      crow_start[c + 1] += crow_start[c];
      cvout_start[c + 1] += cvout_start[c];
    }
    crow.resize(crow_start.back());
    cvout.resize(cvout_start.back());
    std::vector<uint32_t> fr(crow_start.begin(), crow_start.end() - 1), fv(cvout_start.begin(), cvout_start.end() - 1);
    for (size_t ri = 0; ri < rows.size(); ++ri) {
      const NodeId r = rows[ri].root;
      if (r != kNull && comp_of[r] >= 0) crow[fr[comp_of[r]]++] = static_cast<uint32_t>(ri);
    }
    for (size_t v = 0; v < value_outs.size(); ++v) {
      const NodeId r = value_outs[v].node;
      if (r != kNull && comp_of[r] >= 0) cvout[fv[comp_of[r]]++] = static_cast<uint32_t>(v);
    }
  }
  lap("  tape families: components");

  // ---- 2. every component's sequence ----
  std::vector<int32_t>& input_idx = tables.input_idx;
  input_idx.assign(G, -1);
  int32_t n_inputs = 0;
  for (auto& [node, idx] : inputs) {
    input_idx[node] = idx;
    n_inputs = std::max(n_inputs, idx + 1);
  }
  S.n_inputs = n_inputs;
  std::vector<NodeId>& all_nodes = S.all_nodes;  // every component's reachable set, ascending
  std::vector<uint32_t>& all_start = S.all_start;
  all_nodes.clear();
  all_start.assign(1, 0);
  std::vector<uint32_t> seq;               // every component's sequence
  std::vector<uint32_t> seq_start{0};
  std::vector<uint64_t> seq_hash(ncomp);
  // (components are independent of each other: big models on a few threads, each with its own scratch — a stamp and
  // a position per graph node — over a contiguous range of components of about the same number of members)
  struct Chunk {
    size_t c_begin = 0, c_end = 0;
    std::vector<NodeId> all_nodes;
    std::vector<uint32_t> all_count, seq, seq_count;
  };
  // (chunks of about the same number of members, a few per setup thread; a thread's marks — a stamp and a position
  // per graph node — stay with the thread between chunks and between compilations: stamps are handed out from one
  // counter and never repeat)
  const size_t n_chunks = std::max<size_t>(1, std::min<size_t>(members.size() / 16384, 4u * SetupPool::get().threads()));
  std::vector<Chunk> chunks(n_chunks);
  {
    size_t c = 0;
    for (size_t k = 0; k < n_chunks; ++k) {
      chunks[k].c_begin = c;
      const size_t target = members.size() * (k + 1) / n_chunks;
      while (c < ncomp && (k + 1 == n_chunks || comp_start[c + 1] <= target)) ++c;
      chunks[k].c_end = c;
    }
  }
  static std::mutex stamp_mutex;
  static int32_t stamp_next = 0, stamp_generation = 0;
  int32_t stamp_base, generation;
  {
    std::lock_guard<std::mutex> lk(stamp_mutex);
    if (static_cast<int64_t>(stamp_next) + static_cast<int64_t>(ncomp) >= 0x7fffffff) {  // (every thread starts over)
      stamp_next = 0;
      ++stamp_generation;
    }
    stamp_base = stamp_next;
    stamp_next += static_cast<int32_t>(ncomp);
    generation = stamp_generation;
  }
  auto run_chunk = [&](Chunk& ch) {
    const auto t0_ = std::chrono::steady_clock::now();
    static thread_local std::vector<int32_t> stamp;
    static thread_local std::vector<uint32_t> local;
    static thread_local int32_t my_generation = -1;
    if (my_generation != generation) {
      stamp.assign(G, -1);
      my_generation = generation;
    }
    if (stamp.size() < G) stamp.resize(G, -1);
    if (local.size() < G) local.resize(G, 0);
    std::vector<NodeId> ext, stack;
    std::vector<NodeId>& nodes = ch.all_nodes;
    std::vector<uint32_t>& sq = ch.seq;
    const size_t n_mem = comp_start[ch.c_end] - comp_start[ch.c_begin];
    nodes.reserve(n_mem + n_mem / 4);
    sq.reserve(4 * n_mem);
    for (size_t c = ch.c_begin; c < ch.c_end; ++c) {
      ext.clear();
      const int32_t ci = stamp_base + static_cast<int32_t>(c);
      for (uint32_t q = comp_start[c]; q < comp_start[c + 1]; ++q) {
        const NodeId n = members[q];
        for (NodeId a : {g.a0[n], g.a1[n]}) {
          if (a == kNull || comp_of[a] >= 0 || stamp[a] == ci) continue;  // (a member: of this component, by construction)
          stamp[a] = ci;
          ext.push_back(a);
          if (!is_leaf(a)) stack.push_back(a);
          while (!stack.empty()) {  // a private node's own operands: private nodes or leaves
            const NodeId p = stack.back();
            stack.pop_back();
            for (NodeId b : {g.a0[p], g.a1[p]}) {
              if (b == kNull || stamp[b] == ci) continue;
              stamp[b] = ci;
              ext.push_back(b);
              if (!is_leaf(b)) stack.push_back(b);
            }
          }
        }
      }
      std::sort(ext.begin(), ext.end());
      // merge
      const size_t base = nodes.size();
      {
        uint32_t q = comp_start[c];
        size_t e = 0;
        const uint32_t qe = comp_start[c + 1];
        while (q < qe || e < ext.size()) {
          if (e == ext.size() || (q < qe && members[q] < ext[e])) nodes.push_back(members[q++]);
          else nodes.push_back(ext[e++]);
        }
      }
      const size_t cnt = nodes.size() - base;
      for (size_t i = 0; i < cnt; ++i) local[nodes[base + i]] = static_cast<uint32_t>(i);
      ch.all_count.push_back(static_cast<uint32_t>(cnt));
      // sequence
      const size_t s0 = sq.size();
      for (size_t i = 0; i < cnt; ++i) {
        const NodeId n = nodes[base + i];
        uint32_t w0 = g.op[n];
        if (is_leaf(n)) {
          // (what a leaf is bound to: a constant, an input, a parameter — its value or index is the instance's own)
          w0 |= (g.op[n] == OP_CONST ? 1u : (input_idx[n] >= 0 ? 2u : 3u)) << 8;
          sq.push_back(w0);
        } else {
          if (flag[n] & kRepl) w0 |= 1u << 16;
          if (flag[n] & kRoot) w0 |= 1u << 17;
          sq.push_back(w0);
          sq.push_back(local[g.a0[n]]);
          sq.push_back(g.a1[n] == kNull ? 0xffffffffu : local[g.a1[n]]);
        }
      }
      for (uint32_t q = cvout_start[c]; q < cvout_start[c + 1]; ++q) {
        sq.push_back(0xfffffff0u);
        sq.push_back(local[value_outs[cvout[q]].node]);
      }
      for (uint32_t q = crow_start[c]; q < crow_start[c + 1]; ++q) {
        const TapeRow& row = rows[crow[q]];
        sq.push_back(0xfffffff1u);
        sq.push_back(local[row.root]);
        sq.push_back(static_cast<uint32_t>(row.outputs.size()));
        for (auto& o : row.outputs) sq.push_back(stamp[o.wrt] == ci ? local[o.wrt] : 0xffffffffu);
      }
      ch.seq_count.push_back(static_cast<uint32_t>(sq.size() - s0));
      uint64_t h = 1469598103934665603ull;
      for (size_t k = s0; k < sq.size(); ++k) {
        h = (h ^ sq[k]) * 1099511628211ull;
        h ^= h >> 29;
      }
      seq_hash[c] = h;
    }
    if (std::getenv("SLPX_SETUP_TIMING")) std::fprintf(stderr, "    chunk %zu-%zu: %.4f s, %u members, %zu nodes, %zu seq words\n", ch.c_begin, ch.c_end, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(), comp_start[ch.c_end] - comp_start[ch.c_begin], ch.all_nodes.size(), ch.seq.size());
  };
  {
    SetupPool::get().run(static_cast<unsigned>(n_chunks), [&](unsigned k) { run_chunk(chunks[k]); });
    size_t n_all = 0, n_seq = 0;
    for (const Chunk& ch : chunks) {
      n_all += ch.all_nodes.size();
      n_seq += ch.seq.size();
    }
    all_nodes.reserve(n_all);
    seq.reserve(n_seq);
    for (Chunk& ch : chunks) {
      if (n_chunks == 1) {
        all_nodes = std::move(ch.all_nodes);
        seq = std::move(ch.seq);
      } else {
        all_nodes.insert(all_nodes.end(), ch.all_nodes.begin(), ch.all_nodes.end());
        seq.insert(seq.end(), ch.seq.begin(), ch.seq.end());
      }
      for (uint32_t cnt : ch.all_count) all_start.push_back(all_start.back() + cnt);
      for (uint32_t cnt : ch.seq_count) seq_start.push_back(seq_start.back() + cnt);
      ch = Chunk{};
    }
  }
  lap("  tape families: sequences");

  // ---- families: equal sequences ----
  using Family = TapeFamilySet::Family;
  std::vector<Family>& fams = S.fams;
  fams.clear();
  {
    std::unordered_map<uint64_t, std::vector<uint32_t>> by_hash;  // hash -> families with it
    by_hash.reserve(64);
    for (size_t c = 0; c < ncomp; ++c) {
      std::vector<uint32_t>& cand = by_hash[seq_hash[c]];
      const size_t len = seq_start[c + 1] - seq_start[c];
      int64_t hit = -1;
      for (uint32_t f : cand) {
        const uint32_t r = fams[f].rep;
        if (seq_start[r + 1] - seq_start[r] == len &&
            std::equal(seq.begin() + seq_start[c], seq.begin() + seq_start[c + 1], seq.begin() + seq_start[r])) {
          hit = f;
          break;
        }
      }
      if (hit < 0) {
        hit = static_cast<int64_t>(fams.size());
        cand.push_back(static_cast<uint32_t>(fams.size()));
        fams.push_back(Family{static_cast<uint32_t>(c), {}});
      }
      fams[hit].comps.push_back(static_cast<uint32_t>(c));
    }
  }
  // (every parameter leaf the whole program reaches, in node order: flat compiler, "parameters first")
  S.param_order.clear();
  for (size_t n = 0; n < G; ++n)
    if ((flag[n] & kReach) && g.op[n] == OP_VAR && input_idx[n] < 0) S.param_order.push_back(static_cast<NodeId>(n));
  S.accepted.clear();
  S.comp_in_family.assign(ncomp, 0);
  lap("  tape families: classes");
  return true;
}


bool tape_families_accept(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                          const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows,
                          const TapeCompileOptions& opt, TapeFamilySet& S, uint32_t f, const std::vector<uint32_t>& extra_rows) {
  const TapeFamilySet::Family& fam = S.fams[f];
  if (fam.comps.size() < kTapeFamilyMin) return false;
  const uint32_t r = fam.rep;
  if (S.comp_start[r + 1] - S.comp_start[r] < 8) return false;  // (far from the 16 nodes a task of its own takes: not worth a trial)
  FamilyNodeTables& tables = family_node_tables();
  const size_t G = g.size();  // (the caller may have added nodes since the analysis)
  FlatScratch& flat_scratch = tables.flat;
  flat_scratch.bind(G, inputs);
  struct Unbind {
    FlatScratch& f;
    ~Unbind() { f.unbind(); }
  } unbind_flat{flat_scratch};
  std::vector<uint32_t> vsel(S.cvout.begin() + S.cvout_start[r], S.cvout.begin() + S.cvout_start[r + 1]);
  std::vector<uint32_t> rsel(S.crow.begin() + S.crow_start[r], S.crow.begin() + S.crow_start[r + 1]);
  rsel.insert(rsel.end(), extra_rows.begin(), extra_rows.end());
  TapeTrace trace;
  TapeProgram rp = compile_tape_flat(g, inputs, value_outs, rows, vsel, rsel, opt, &trace, /*tail_padding=*/false, nullptr, &flat_scratch);
  // (what the flat compiler gives a family member: a task of its own — from 16 nodes)
  if (rp.tasks.size() != 1 || trace.tasks.size() != 1 || trace.tasks[0].n_comps != 1 || trace.tasks[0].comp_nodes < 16) return false;
  // every leaf must be findable by position in the member's reachable set — or be the caller's to translate
  // (a leaf the caller's extra rows brought in: beyond the analysed graph, or a multiplier: not reachable from the
  // component's own roots)
  if (extra_rows.empty()) {
    std::vector<uint32_t>& local = tables.local;
    if (local.size() < G) local.resize(G, 0);
    bool ok = true;
    for (uint32_t q = S.all_start[r]; q < S.all_start[r + 1]; ++q) local[S.all_nodes[q]] = q - S.all_start[r];
    for (NodeId n : trace.tasks[0].leaf_nodes)
      ok = ok && n != kNull && local[n] < S.all_start[r + 1] - S.all_start[r] && S.all_nodes[S.all_start[r] + local[n]] == n;
    if (!ok) return false;
  }
  for (uint32_t c : fam.comps) S.comp_in_family[c] = 1;
  S.accepted.push_back(TapeFamilySet::Accepted{f, std::move(rp), std::move(trace.tasks[0]), extra_rows});
  return true;
}

TapeProgram tape_families_emit(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                               const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows,
                               const TapeCompileOptions& opt, TapeFamilySet& S, const TapeFamilyHooks& hooks,
                               const std::vector<uint32_t>& more_vouts, const std::vector<uint32_t>& more_rows) {
  SetupLap lap;
  TapeProgram out;
  FamilyNodeTables& tables = family_node_tables();
  const size_t G = g.size();  // (the caller may have added nodes since the analysis)
  const size_t ncomp = S.ncomp;
  const std::vector<uint32_t>&comp_start = S.comp_start, &crow_start = S.crow_start, &cvout_start = S.cvout_start, &crow = S.crow,
                             &cvout = S.cvout, &loose_rows = S.loose_rows, &loose_vouts = S.loose_vouts, &all_start = S.all_start;
  (void)comp_start;
  const std::vector<NodeId>& all_nodes = S.all_nodes;
  const std::vector<TapeFamilySet::Family>& fams = S.fams;
  std::vector<TapeFamilySet::Accepted>& accepted = S.accepted;
  const std::vector<uint8_t>& comp_in_family = S.comp_in_family;
  const int32_t n_inputs = S.n_inputs;
  // (another analysis on this thread may have used the tables since: the inputs again)
  std::vector<int32_t>& input_idx = tables.input_idx;
  input_idx.assign(G, -1);
  for (auto& [node, idx] : inputs) input_idx[node] = idx;
  std::vector<uint32_t>& local = tables.local;  // (scratch of the passes below)
  if (local.size() < G) local.resize(G, 0);
  FlatScratch& flat_scratch = tables.flat;
  flat_scratch.bind(G, inputs);
  struct Unbind {
    FlatScratch& f;
    ~Unbind() { f.unbind(); }
  } unbind_flat{flat_scratch};
  // ---- 4. everything else through the flat compiler together ----
  {
    std::vector<uint32_t> vsel = loose_vouts, rsel = loose_rows;
    for (size_t c = 0; c < ncomp; ++c) {
      if (comp_in_family[c]) continue;
      vsel.insert(vsel.end(), cvout.begin() + cvout_start[c], cvout.begin() + cvout_start[c + 1]);
      rsel.insert(rsel.end(), crow.begin() + crow_start[c], crow.begin() + crow_start[c + 1]);
    }
    vsel.insert(vsel.end(), more_vouts.begin(), more_vouts.end());
    rsel.insert(rsel.end(), more_rows.begin(), more_rows.end());
    std::sort(vsel.begin(), vsel.end());
    std::sort(rsel.begin(), rsel.end());
    out = compile_tape_flat(g, inputs, value_outs, rows, vsel, rsel, opt, nullptr, /*tail_padding=*/false, &S.param_order, &flat_scratch);
  }
  lap("  tape families: representatives + remainder");
  TapeProgram& prog = out;
  prog.n_inputs = std::max(prog.n_inputs, n_inputs);
  // constants: one pool for the whole program (a parameter's slot is never shared with a literal)
  std::unordered_map<double, uint32_t> const_pool;
  std::unordered_map<NodeId, uint32_t> param_slot;
  {
    std::vector<uint8_t> is_param(prog.consts.size(), 0);
    for (auto& [node, slot] : prog.params) {
      is_param[slot] = 1;
      param_slot.emplace(node, slot);
    }
    for (size_t i = 0; i < prog.consts.size(); ++i)
      if (!is_param[i]) const_pool.emplace(prog.consts[i], static_cast<uint32_t>(i));
  }
  auto leaf_binding = [&](NodeId n) -> uint32_t {
    if (g.op[n] == OP_CONST) {
      auto it = const_pool.find(g.val[n]);
      if (it == const_pool.end()) {
        it = const_pool.emplace(g.val[n], static_cast<uint32_t>(prog.consts.size())).first;
        prog.consts.push_back(g.val[n]);
      }
      return kLeafConstFlag | it->second;
    }
    if (input_idx[n] >= 0) return static_cast<uint32_t>(input_idx[n]);
    auto pit = param_slot.find(n);
    if (pit == param_slot.end()) {
      const uint32_t i = static_cast<uint32_t>(prog.consts.size());
      prog.consts.push_back(g.val[n]);
      prog.params.emplace_back(n, i);
      pit = param_slot.emplace(n, i).first;
    }
    return kLeafConstFlag | pit->second;
  };
  for (TapeFamilySet::Accepted& acc : accepted) {
    const TapeFamilySet::Family& fam = fams[acc.fam];
    const TapeProgram& rp = acc.prog;
    const uint32_t r = fam.rep;
    // the family's structure, once
    while ((prog.node_rec.size() / 3) % 2) {
      for (int k = 0; k < 3; ++k) prog.node_rec.push_back(0);
      for (int k = 0; k < 4; ++k) prog.node_rec16.push_back(0);
    }
    while (prog.edges.size() % 4) {
      prog.edges.push_back({0, 0});
      prog.edges16.push_back(0);
      prog.edges16.push_back(0);
    }
    while (prog.slot_edge_ptr.size() % 8) {
      prog.slot_edge_ptr.push_back(0);
      prog.slot_edge_ptr16.push_back(0);
    }
    while (prog.lvl_ptr.size() % 4) prog.lvl_ptr.push_back(0);
    while (prog.slvl_ptr.size() % 4) prog.slvl_ptr.push_back(0);
    TapeTask base = rp.tasks[0];
    base.node_off += static_cast<uint32_t>(prog.node_rec.size() / 3);
    base.lvl_off += static_cast<uint32_t>(prog.lvl_ptr.size());
    base.slot_off += static_cast<uint32_t>(prog.slot_edge_ptr.size());
    base.slvl_off += static_cast<uint32_t>(prog.slvl_ptr.size());
    base.edge_off += static_cast<uint32_t>(prog.edges.size());
    prog.node_rec.insert(prog.node_rec.end(), rp.node_rec.begin(), rp.node_rec.end());
    prog.node_rec16.insert(prog.node_rec16.end(), rp.node_rec16.begin(), rp.node_rec16.end());
    prog.lvl_ptr.insert(prog.lvl_ptr.end(), rp.lvl_ptr.begin(), rp.lvl_ptr.end());
    prog.slot_edge_ptr.insert(prog.slot_edge_ptr.end(), rp.slot_edge_ptr.begin(), rp.slot_edge_ptr.end());
    prog.slot_edge_ptr16.insert(prog.slot_edge_ptr16.end(), rp.slot_edge_ptr16.begin(), rp.slot_edge_ptr16.end());
    prog.slvl_ptr.insert(prog.slvl_ptr.end(), rp.slvl_ptr.begin(), rp.slvl_ptr.end());
    prog.edges.insert(prog.edges.end(), rp.edges.begin(), rp.edges.end());
    prog.edges16.insert(prog.edges16.end(), rp.edges16.begin(), rp.edges16.end());
    prog.basic_ops = prog.basic_ops && rp.basic_ops;
    prog.max_levels = std::max(prog.max_levels, rp.max_levels);
    prog.max_slot_levels = std::max(prog.max_slot_levels, rp.max_slot_levels);
    // positions, in the member's reachable set / row list / value-output list, of what the trace names
    const TapeTrace::Task& tt = acc.tt;
    for (uint32_t q = all_start[r]; q < all_start[r + 1]; ++q) local[all_nodes[q]] = q - all_start[r];
    std::vector<uint32_t> leaf_pos(tt.leaf_nodes.size()), vout_pos(tt.vouts.size());
    std::vector<std::pair<uint32_t, uint32_t>> jout_pos(tt.jouts.size());  // (row of the member, output of the row)
    bool ok = true;
    constexpr uint32_t kOutside = 0xffffffffu;  // a leaf that is not of the component: the caller's to translate
    const uint32_t n_all_r = all_start[r + 1] - all_start[r];
    for (size_t i = 0; i < tt.leaf_nodes.size(); ++i) {
      const NodeId ln = tt.leaf_nodes[i];
      const bool inside = static_cast<size_t>(ln) < S.graph_size && local[ln] < n_all_r && all_nodes[all_start[r] + local[ln]] == ln;
      leaf_pos[i] = inside ? local[ln] : kOutside;
      if (!inside && !hooks.outside_leaf) ok = false;
    }
    for (size_t i = 0; i < tt.vouts.size(); ++i) {
      const auto b = cvout.begin() + cvout_start[r], e = cvout.begin() + cvout_start[r + 1];
      const auto it = std::find(b, e, tt.vouts[i]);
      ok = ok && it != e;
      vout_pos[i] = static_cast<uint32_t>(it - b);
    }
    constexpr uint32_t kExtraRow = 0x80000000u;  // jout_pos.first: | position in the family's extra rows
    for (size_t i = 0; i < tt.jouts.size(); ++i) {
      const auto b = crow.begin() + crow_start[r], e = crow.begin() + crow_start[r + 1];
      const auto it = std::find(b, e, tt.jouts[i].first);
      uint32_t where;
      const TapeRow* row;
      if (it != e) {
        where = static_cast<uint32_t>(it - b);
        row = &rows[*it];
      } else {
        const auto xit = std::find(acc.extra_rows.begin(), acc.extra_rows.end(), tt.jouts[i].first);
        ok = ok && xit != acc.extra_rows.end() && static_cast<bool>(hooks.extra_dst);
        if (!ok) break;
        where = kExtraRow | static_cast<uint32_t>(xit - acc.extra_rows.begin());
        row = &rows[*xit];
      }
      size_t j = 0;
      while (j < row->outputs.size() && row->outputs[j].wrt != tt.jouts[i].second) ++j;
      ok = ok && j < row->outputs.size();
      jout_pos[i] = {where, static_cast<uint32_t>(j)};
    }
    if (!ok) throw std::runtime_error("slpx tape compiler: a family representative's trace does not match its component");
    const int cls = tt.cls;
    for (uint32_t c : fam.comps) {
      while (prog.leaf_src.size() % 4) prog.leaf_src.push_back(0);
      TapeTask t = base;
      t.leaf_off = static_cast<uint32_t>(prog.leaf_src.size());
      t.vout_off = static_cast<uint32_t>(prog.vout_src.size());
      t.jout_off = static_cast<uint32_t>(prog.jout_slot.size());
      const NodeId* nodes_c = all_nodes.data() + all_start[c];
      for (size_t i = 0; i < leaf_pos.size(); ++i)
        prog.leaf_src.push_back(leaf_binding(leaf_pos[i] != kOutside ? nodes_c[leaf_pos[i]] : hooks.outside_leaf(acc.fam, tt.leaf_nodes[i], c)));
      for (size_t i = 0; i < vout_pos.size(); ++i) {
        const TapeValueOut& vo = value_outs[cvout[cvout_start[c] + vout_pos[i]]];
        prog.vout_src.push_back(rp.vout_src[rp.tasks[0].vout_off + i]);
        prog.vout_dst.push_back(static_cast<uint32_t>(vo.dst));
        prog.vout_scale.push_back(vo.scale_idx);
      }
      for (size_t i = 0; i < jout_pos.size(); ++i) {
        prog.jout_slot.push_back(rp.jout_slot[rp.tasks[0].jout_off + i]);
        if (jout_pos[i].first & kExtraRow) {
          const uint32_t k = jout_pos[i].first & ~kExtraRow;
          prog.jout_dst.push_back(static_cast<uint32_t>(hooks.extra_dst(acc.fam, k, jout_pos[i].second, c)));
          prog.jout_scale.push_back(rows[acc.extra_rows[k]].scale_idx);
        } else {
          const TapeRow& row = rows[crow[crow_start[c] + jout_pos[i].first]];
          prog.jout_dst.push_back(static_cast<uint32_t>(row.outputs[jout_pos[i].second].dst));
          prog.jout_scale.push_back(row.scale_idx);
        }
      }
      const uint32_t ti = static_cast<uint32_t>(prog.tasks.size());
      if (cls == 0) {
        prog.small_tasks.push_back(ti);
        prog.small_lds_bytes = std::max(prog.small_lds_bytes, t.lds_bytes);
      } else if (cls == 1) {
        prog.large_tasks.push_back(ti);
        prog.large_lds_bytes = std::max(prog.large_lds_bytes, t.lds_bytes);
      } else {
        prog.global_tasks.push_back(ti);
        t.scratch_off = static_cast<uint32_t>(prog.global_scratch_doubles);
        prog.global_scratch_doubles += t.lds_doubles;
      }
      prog.tasks.push_back(t);
      prog.total_nodes += t.n_node;
      prog.total_slots += t.n_slot;
      prog.total_leaves += t.n_leaf;
      prog.total_edges += t.n_edge;
      if (c != r) ++prog.shared_tasks;
    }
  }
  add_tail_padding(prog);
  lap("  tape families: instances");
  return out;
}



// The whole of it for a caller with nothing to do between the phases: every family as it stands.
static bool compile_tape_families(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                                  const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows,
                                  const TapeCompileOptions& opt, TapeProgram& out) {
  TapeFamilySet S;
  if (!tape_families_analyze(g, inputs, value_outs, rows, S)) return false;
  for (size_t f = 0; f < S.fams.size(); ++f) tape_families_accept(g, inputs, value_outs, rows, opt, S, static_cast<uint32_t>(f), {});
  if (S.accepted.empty()) return false;
  out = tape_families_emit(g, inputs, value_outs, rows, opt, S, TapeFamilyHooks{});
  return true;
}

TapeProgram compile_tape(Graph& g, const std::vector<std::pair<NodeId, int32_t>>& inputs,
                         const std::vector<TapeValueOut>& value_outs, const std::vector<TapeRow>& rows,
                         const TapeCompileOptions& opt) {
  bool families = opt.families;
  if (const char* env = std::getenv("SLPX_TAPE_TEMPLATES")) families = env[0] != '0';
  TapeProgram prog;
  bool done = false;
  if (families && value_outs.size() + rows.size() >= 64) done = compile_tape_families(g, inputs, value_outs, rows, opt, prog);
  if (!done) {
    std::vector<uint32_t> vsel(value_outs.size()), rsel(rows.size());
    std::iota(vsel.begin(), vsel.end(), 0u);
    std::iota(rsel.begin(), rsel.end(), 0u);
    prog = compile_tape_flat(g, inputs, value_outs, rows, vsel, rsel, opt, nullptr, /*tail_padding=*/true);
  }
  for (auto& v : value_outs) prog.n_outputs = std::max(prog.n_outputs, v.dst + 1);
  for (auto& r : rows)
    for (auto& o : r.outputs) prog.n_outputs = std::max(prog.n_outputs, o.dst + 1);
  return prog;
}

}  // namespace slpx
