#include "nlp.hpp"

#include "setup_timing.hpp"
#include "setup_threads.hpp"

#include <unordered_map>

#include <algorithm>
#include <future>
#include <functional>
#include <memory>
#include <stdexcept>

namespace slpx {

namespace {

struct RowEntry {
  int32_t row, col;
  NodeId wrt;
  double cached;  // value when the row is LINEAR
  bool is_cached;
};

// One derivative matrix (the body of Jacobian's constructor, jacobian.hpp:54-105):
// per-row parent->child lists, (col,node) output lists, LINEAR rows evaluated once.
struct MatrixBuild {
  CscPattern pat;
  std::vector<RowEntry> entries;       // in CSC order after finalize()
  std::vector<int32_t> nonlinear_rows;
  int linear_rows = 0;
};

// Adjoint sweep of a LINEAR row on the host.  Only + - neg, const*x and x/const
// can appear in a LINEAR expression (expression.hpp:155-348), and the formulas
// below are the reference's grad_l/grad_r for exactly those ops
// (expression.hpp:444-515, 616-694, 696-730).
void linear_row_adjoints(Graph& g, const std::vector<NodeId>& top, double* adj) {
  g.update_values(top);
  for (NodeId n : top) adj[n] = 0.0;
  adj[top[0]] = 1.0;
  for (NodeId n : top) {
    NodeId l = g.a0[n], r = g.a1[n];
    if (l == kNull) continue;
    const double a = adj[n];
    switch (static_cast<Opcode>(g.op[n])) {
      case OP_ADD: adj[l] += a; adj[r] += a; break;
      case OP_SUB: adj[l] += a; adj[r] += -a; break;
      case OP_NEG: adj[l] += -a; break;
      case OP_MUL: adj[l] += a * g.val[r]; adj[r] += a * g.val[l]; break;
      case OP_DIV:
        adj[l] += a / g.val[r];
        adj[r] += a * -g.val[l] / (g.val[r] * g.val[r]);
        break;
      default:
        throw std::runtime_error("linear_row_adjoints: non-linear op in a LINEAR row");
    }
  }
}

// Row visits of build_matrix: a stamp per graph node (which row saw it last) and a stack, shared by the five matrices.
struct RowVisit {
  std::vector<int32_t> col;  // the column of a wrt node, -1 otherwise (Graph::scratch is topological_sort's)
  std::unique_ptr<double[]> adj;
  size_t adj_size = 0;
};

MatrixBuild build_matrix(Graph& g, const std::vector<NodeId>& rows, const std::vector<NodeId>& wrt,
                         int nrows, int ncols, bool lower, RowVisit& visit) {
  MatrixBuild mb;
  // Like the reference, tag each wrt node with its column (jacobian.hpp:64-66).  A LINEAR row needs its
  // parent->child list (topological_sort; the order fixes how its constant adjoints are summed); of a nonlinear row
  // only the SET of wrt nodes it reaches is needed here (the tape compiler makes its own lists): one marking walk,
  // every node of the row touched once — the lists of all rows of a Hessian were 1.2 million entries at N=1000.
  // adjoints of a LINEAR row's nodes: written before they are read (linear_row_adjoints), so the buffer is left
  // as it comes — a zeroed vector of a million doubles was 2 ms of page faults for a matrix with one row
  if (visit.adj_size < g.size()) {
    visit.adj.reset(new double[g.size()]);
    visit.adj_size = g.size();
  }
  double* adj_scratch = visit.adj.get();
  // (the table reaches as far as the last wrt node — the decision variables are among a model's first nodes —
  // not over the whole graph: `col_at` answers -1 beyond it)
  NodeId last_wrt = -1;
  for (NodeId w : wrt) last_wrt = std::max(last_wrt, w);
  if (visit.col.size() < static_cast<size_t>(last_wrt) + 1) visit.col.resize(static_cast<size_t>(last_wrt) + 1, -1);
  struct ColOf {
    std::vector<int32_t>& v;
    NodeId last;
    int32_t operator[](NodeId n) const { return n <= last ? v[n] : -1; }
  } col_of{visit.col, last_wrt};
  for (size_t c = 0; c < wrt.size(); ++c) visit.col[wrt[c]] = static_cast<int32_t>(c);
  // (blank again however the function is left: a walk that throws on a pool thread must not leave the next model
  // built on this thread with this one's columns)
  struct BlankCols {
    std::vector<int32_t>& col;
    const std::vector<NodeId>& wrt;
    ~BlankCols() {
      for (NodeId w : wrt) col[w] = -1;
    }
  } blank_cols{visit.col, wrt};
  // the rows in chunks on the setup threads (a walk only reads the graph; every chunk has its own marks): the
  // entries of a chunk in row order, the chunks one after the other — the order a single thread finds them in
  struct Chunk {
    std::vector<RowEntry> entries;
    std::vector<int32_t> nonlinear_rows;
    int linear_rows = 0;
  };
  const unsigned n_chunks = std::max(1u, parallel_chunk_count(rows.size(), 256));
  std::vector<Chunk> chunks(n_chunks);
  // LINEAR rows first, on this thread: their sort uses the graph's own scratch marks and refreshes node values
  std::vector<std::vector<RowEntry>> linear_entries;  // per linear row, in row order
  std::vector<int32_t> linear_row_index;
  for (size_t r = 0; r < rows.size(); ++r) {
    if (rows[r] == kNull || g.type[rows[r]] != T_LINEAR) continue;
    linear_row_index.push_back(static_cast<int32_t>(r));
    linear_entries.emplace_back();
    const std::vector<NodeId> top = g.topological_sort(rows[r]);
    if (top.empty()) continue;
    linear_row_adjoints(g, top, adj_scratch);
    for (NodeId n : top) {
      const int32_t col = col_of[n];
      if (col == -1 || (lower && col > static_cast<int32_t>(r))) continue;
      linear_entries.back().push_back({static_cast<int32_t>(r), col, n, adj_scratch[n], true});
    }
  }
  const size_t graph_size = g.size();
  parallel_chunks(rows.size(), 256, [&](size_t r_begin, size_t r_end, unsigned ci) {
    Chunk& ch = chunks[ci];
    // marks of this THREAD (whichever chunks it gets): stamps only grow, so marks left by earlier rows, matrices
    // or graphs never match
    // (a thread keeps its marks between rows, matrices and models — up to a few million nodes; a graph beyond that
    // gets them for the call only: sixteen pool threads holding marks of a six-million-node graph for the life of the
    // process were hundreds of MB)
    static thread_local std::vector<int32_t> seen;
    static thread_local int32_t stamp = 0;
    struct ReleaseBig {
      std::vector<int32_t>& v;
      ~ReleaseBig() {
        if (v.size() > (4u << 20)) std::vector<int32_t>().swap(v);
      }
    } release_big{seen};
    std::vector<NodeId> stack;
    std::vector<std::pair<int32_t, NodeId>> outs;
    size_t lin = static_cast<size_t>(std::lower_bound(linear_row_index.begin(), linear_row_index.end(), static_cast<int32_t>(r_begin)) - linear_row_index.begin());
    for (size_t r = r_begin; r < r_end; ++r) {
      if (rows[r] == kNull) continue;
      const uint8_t t = g.type[rows[r]];
      if (t == T_LINEAR) {
        ++ch.linear_rows;
        ch.entries.insert(ch.entries.end(), linear_entries[lin].begin(), linear_entries[lin].end());
        ++lin;
      } else if (t > T_LINEAR) {
        ch.nonlinear_rows.push_back(static_cast<int32_t>(r));
        if (seen.size() < graph_size) seen.resize(graph_size, -1);
        const int32_t st = stamp++;
        stack.assign(1, rows[r]);
        seen[rows[r]] = st;
        outs.clear();
        while (!stack.empty()) {
          const NodeId n = stack.back();
          stack.pop_back();
          const NodeId l = g.a0[n], rr = g.a1[n];
          if (l == kNull) {
            if (col_of[n] != -1) outs.emplace_back(col_of[n], n);
            continue;
          }
          if (seen[l] != st) {
            seen[l] = st;
            stack.push_back(l);
          }
          if (rr != kNull && seen[rr] != st) {
            seen[rr] = st;
            stack.push_back(rr);
          }
        }
        for (auto& [col, node] : outs) {
          if (lower && col > static_cast<int32_t>(r)) continue;
          ch.entries.push_back({static_cast<int32_t>(r), col, node, 0.0, false});
        }
      }
    }
  });
  // CSC order (setFromTriplets: column-major, rows ascending): the entries come row by row, so a stable
  // counting sort by column leaves the rows of a column ascending
  mb.pat.rows = nrows;
  mb.pat.cols = ncols;
  mb.pat.colptr.assign(ncols + 1, 0);
  size_t total = 0;
  for (const Chunk& ch : chunks) {
    total += ch.entries.size();
    for (const RowEntry& e : ch.entries) ++mb.pat.colptr[e.col + 1];
    mb.nonlinear_rows.insert(mb.nonlinear_rows.end(), ch.nonlinear_rows.begin(), ch.nonlinear_rows.end());
    mb.linear_rows += ch.linear_rows;
  }
  for (int c = 0; c < ncols; ++c) mb.pat.colptr[c + 1] += mb.pat.colptr[c];
  mb.entries.resize(total);
  mb.pat.rowidx.resize(total);
  {
    std::vector<int32_t> next(mb.pat.colptr.begin(), mb.pat.colptr.end() - 1);
    for (const Chunk& ch : chunks)
      for (const RowEntry& e : ch.entries) {
        const int32_t q = next[e.col]++;
        mb.entries[q] = e;
        mb.pat.rowidx[q] = e.row;
      }
  }
  return mb;
}

}  // namespace

NlpStructure build_nlp_structure(Graph& g, const std::vector<NodeId>& x, NodeId f_in,
                                 const std::vector<NodeId>& c_e, const std::vector<NodeId>& c_i,
                                 const TapeCompileOptions& opt,
                                 const std::function<void(const NlpStructure&)>& on_patterns) {
  NlpStructure s;
  SetupLap lap;
  s.graph_nodes_before = g.size();
  s.n = static_cast<int>(x.size());
  s.m_e = static_cast<int>(c_e.size());
  s.m_i = static_cast<int>(c_i.size());
  const int n = s.n, m_e = s.m_e, m_i = s.m_i;

  // problem.hpp:236-263
  s.f_type = f_in == kNull ? T_NONE : g.type[f_in];
  for (NodeId c : c_e) s.ce_type = std::max(s.ce_type, g.type[c]);
  for (NodeId c : c_i) s.ci_type = std::max(s.ci_type, g.type[c]);

  NodeId f = f_in == kNull ? g.constant(0.0) : f_in;  // problem.hpp:318

  // (the gradient trees below add about twice the model's own nodes: one allocation instead of a dozen doublings)
  g.reserve(g.size() + 2 * g.size() + 4 * static_cast<size_t>(m_e + m_i) + 1024);
  // problem.hpp:519-520: dual variables as fresh decision-variable leaves
  for (int j = 0; j < m_e; ++j) s.y_nodes.push_back(g.variable(0.0));
  for (int j = 0; j < m_i; ++j) s.z_nodes.push_back(g.variable(0.0));

  // problem.hpp:542: H_f rows = symbolic gradient of f
  std::vector<NodeId> Hf_rows = g.gradient_tree(g.topological_sort(f), x);
  lap("gradient tree of the cost (H_f rows)");

  // (kept by the thread between models: build_matrix leaves `col` blank again)
  static thread_local RowVisit visit;
  SetupLap lap_m;
  MatrixBuild mg = build_matrix(g, {f}, x, 1, n, false, visit);        // problem.hpp:535
  lap_m("  rows: g");
  MatrixBuild mHf = build_matrix(g, Hf_rows, x, n, n, true, visit);
  lap_m("  rows: H_f");
  MatrixBuild mAe = build_matrix(g, c_e, x, m_e, n, false, visit);     // problem.hpp:555
  lap_m("  rows: A_e");
  MatrixBuild mAi = build_matrix(g, c_i, x, m_i, n, false, visit);     // problem.hpp:560
  lap_m("  rows: A_i");
  lap("row lists + patterns (g, H_f, A_e, A_i)");

  // V = [f | c_e | c_i | g | A_e | A_i | H_f | H_c | partial sums]: everything in front of the H_c block is placed
  // now, the H_c block — the pattern that is still to come — and what lies behind it in finish_layout
  s.g_pat = mg.pat;
  s.Ae = mAe.pat;
  s.Ai = mAi.pat;
  s.Hf = mHf.pat;
  s.off_f = 0;
  s.off_ce = 1;
  s.off_ci = s.off_ce + m_e;
  s.off_g = s.off_ci + m_i;
  s.off_Ae = s.off_g + s.g_pat.nnz();
  s.off_Ai = s.off_Ae + s.Ae.nnz();
  s.off_Hf = s.off_Ai + s.Ai.nnz();
  s.off_Hc = s.off_Hf + s.Hf.nnz();
  s.nV = s.off_Hc;
  s.V_static_raw.assign(s.nV, 0.0);
  s.V_scale_idx.assign(s.nV, -1);
  s.V_is_static.assign(s.nV, 1);
  constexpr int32_t kTailBase = 1 << 30;  // destinations in the hidden tail, until its place is known
  int32_t n_tail = 0;
  std::vector<double> tail_static;
  std::vector<uint8_t> tail_is_static;

  std::vector<std::pair<NodeId, int32_t>> inputs;
  for (int i = 0; i < n; ++i) inputs.emplace_back(x[i], i);
  for (int j = 0; j < m_e; ++j) inputs.emplace_back(s.y_nodes[j], n + j);
  for (int j = 0; j < m_i; ++j) inputs.emplace_back(s.z_nodes[j], n + m_e + j);

  std::vector<TapeValueOut> vouts;
  vouts.push_back({f, s.off_f, 0});
  for (int j = 0; j < m_e; ++j) vouts.push_back({c_e[j], s.off_ce + j, 1 + j});
  for (int j = 0; j < m_i; ++j) vouts.push_back({c_i[j], s.off_ci + j, 1 + m_e + j});
  for (auto& v : vouts) {
    s.V_scale_idx[v.dst] = v.scale_idx;
    s.V_is_static[v.dst] = 0;
  }
  // A CONSTANT value root is never touched by a sweep; bake it in
  std::vector<TapeValueOut> live_vouts;
  for (auto& v : vouts) {
    if (g.type[v.node] == T_CONSTANT) {
      s.V_static_raw[v.dst] = g.val[v.node];
      s.V_is_static[v.dst] = 1;
    } else {
      live_vouts.push_back(v);
    }
  }

  std::vector<TapeRow> rows;
  // `slot_of`: where entry k of the matrix lies in V (the identity for everything but a merged H_c)
  auto add_matrix = [&](MatrixBuild& mb, const std::vector<NodeId>& roots, int off, const std::function<int32_t(int32_t)>& scale_of_row,
                        const std::vector<int32_t>* slot_of = nullptr) {
    std::vector<TapeRow> mrows(roots.size());
    for (size_t k = 0; k < mb.entries.size(); ++k) {
      const RowEntry& e = mb.entries[k];
      const int32_t dst = off + (slot_of ? (*slot_of)[k] : static_cast<int32_t>(k));
      s.V_scale_idx[dst] = scale_of_row(e.row);
      if (e.is_cached) {
        s.V_static_raw[dst] = e.cached;
      } else {
        s.V_is_static[dst] = 0;
        mrows[e.row].outputs.push_back({e.wrt, dst});
      }
    }
    for (int32_t r : mb.nonlinear_rows) {
      if (mrows[r].outputs.empty()) continue;  // e.g. every entry above the diagonal
      mrows[r].root = roots[r];
      mrows[r].scale_idx = scale_of_row(r);
      rows.push_back(std::move(mrows[r]));
    }
    s.nonlinear_rows += static_cast<int>(mb.nonlinear_rows.size());
    s.linear_rows += mb.linear_rows;
  };
  lap("  V layout: offsets, value outputs");
  add_matrix(mg, {f}, s.off_g, [](int32_t) { return 0; });
  add_matrix(mAe, c_e, s.off_Ae, [](int32_t r) { return 1 + r; });
  add_matrix(mAi, c_i, s.off_Ai, [m_e](int32_t r) { return 1 + m_e + r; });
  add_matrix(mHf, Hf_rows, s.off_Hf, [](int32_t) { return 0; });
  lap("  V layout: matrices");
  // ---- long separable sums ----------------------------------------------------------
  // A cost like sum_k u_k^2 is ONE connected component (its ADD tree) however independent
  // its terms are: at N=1000 a 152 KB-LDS task, at N=5000 one that only fits in HBM scratch,
  // both on the critical path of every sweep.  When the terms of the sum touch disjoint
  // sets of decision variables the sum is cut into groups of consecutive terms: every group
  // becomes a small component of its own (a template group, usually) that writes its partial
  // sum into a hidden tail of V and differentiates ITS terms (d f / d partial = 1), and a
  // tiny reduce kernel adds the partials in a fixed order.
  if (f_in != kNull && opt.split_sum_min_terms > 0) {
    // the chain of additions under f: f = ((t_0 + t_1) + t_2) + ...; a link of it used by anything else ends it.
    // (Only the links' use counts matter: the candidates are marked, and the whole graph — six million nodes at
    // N=5000 — is scanned for references to marked nodes only, in chunks on the setup threads.)
    std::vector<NodeId> links;
    for (NodeId cur = f; g.op[cur] == OP_ADD; cur = g.a0[cur]) links.push_back(cur);
    std::vector<int32_t> link_of(static_cast<size_t>(f) + 1, -1);
    for (size_t i = 0; i < links.size(); ++i) link_of[links[i]] = static_cast<int32_t>(i);
    std::vector<int32_t> link_uses(links.size(), 0);
    if (!links.empty()) {
      const unsigned n_chunks = std::max(1u, parallel_chunk_count(g.size(), 1u << 16));
      std::vector<std::vector<int32_t>> hits(n_chunks);
      parallel_chunks(g.size(), 1u << 16, [&](size_t b, size_t e, unsigned ci) {
        for (size_t k = b; k < e; ++k) {
          const NodeId l = g.a0[k], r = g.a1[k];
          if (l != kNull && l <= f && link_of[l] >= 0) hits[ci].push_back(link_of[l]);
          if (r != kNull && r <= f && link_of[r] >= 0) hits[ci].push_back(link_of[r]);
        }
      });
      for (auto& h : hits)
        for (int32_t i : h) ++link_uses[i];
    }
    std::vector<NodeId> terms;
    NodeId cur = f;
    while (g.op[cur] == OP_ADD && (cur == f || link_uses[link_of[cur]] == 1)) {
      terms.push_back(g.a1[cur]);
      cur = g.a0[cur];
    }
    terms.push_back(cur);
    std::reverse(terms.begin(), terms.end());
    if (terms.size() >= opt.split_sum_min_terms) {
      // (everything below f has a smaller number than f)
      std::vector<int32_t> xindex_of(static_cast<size_t>(f) + 1, -1);
      for (int i = 0; i < n; ++i)
        if (x[i] <= f) xindex_of[x[i]] = i;
      struct {
        const std::vector<int32_t>& v;
        int32_t at(NodeId node) const {
          if (static_cast<size_t>(node) >= v.size() || v[node] < 0) throw std::runtime_error("slpx: a gradient entry of the cost outside its variables");
          return v[node];
        }
      } xindex{xindex_of};
      const size_t gs = opt.split_sum_group;
      const size_t G = (terms.size() + gs - 1) / gs;
      std::vector<int32_t> group_of_var(n, -1);
      std::vector<int32_t> stamp(static_cast<size_t>(f) + 1, -1);
      bool separable = true;
      std::vector<NodeId> stack;
      for (size_t t = 0; t < terms.size() && separable; ++t) {
        const int32_t grp = static_cast<int32_t>(t / gs);
        stack.assign(1, terms[t]);
        while (!stack.empty() && separable) {
          const NodeId v = stack.back();
          stack.pop_back();
          if (stamp[v] == static_cast<int32_t>(t)) continue;
          stamp[v] = static_cast<int32_t>(t);
          const int32_t xi = xindex_of[v];
          if (xi >= 0) {
            if (group_of_var[xi] >= 0 && group_of_var[xi] != grp) separable = false;
            group_of_var[xi] = grp;
          }
          if (g.a0[v] != kNull) stack.push_back(g.a0[v]);
          if (g.a1[v] != kNull) stack.push_back(g.a1[v]);
        }
      }
      if (separable) {
        std::vector<NodeId> partial(G);
        for (size_t j = 0; j < G; ++j) {
          std::vector<NodeId> level(terms.begin() + j * gs, terms.begin() + std::min(terms.size(), (j + 1) * gs));
          while (level.size() > 1) {  // pairwise tree, left-to-right term order
            std::vector<NodeId> next;
            for (size_t k = 0; k + 1 < level.size(); k += 2) next.push_back(g.add(level[k], level[k + 1]));
            if (level.size() & 1) next.push_back(level.back());
            level.swap(next);
          }
          partial[j] = level[0];
        }
        // (the hidden tail of V lies behind the H_c block, whose size is not known yet: placed in finish_layout)
        const int32_t off_part = kTailBase;
        n_tail = static_cast<int32_t>(G);
        tail_static.assign(G, 0.0);
        tail_is_static.assign(G, 0);
        s.reduces.push_back({s.off_f, 0, off_part, static_cast<int32_t>(G)});
        std::vector<TapeValueOut> v2;
        for (auto& v : live_vouts)
          if (v.node != f) v2.push_back(v);
        for (size_t j = 0; j < G; ++j) {
          if (g.type[partial[j]] == T_CONSTANT) {
            tail_static[j] = g.val[partial[j]];
            tail_is_static[j] = 1;
          } else {
            v2.push_back({partial[j], off_part + static_cast<int32_t>(j), -1});
          }
        }
        live_vouts.swap(v2);
        std::vector<TapeRow> r2;
        for (auto& row : rows) {
          if (row.root != f) {
            r2.push_back(std::move(row));
            continue;
          }
          std::vector<TapeRow> parts(G);
          for (auto& o : row.outputs) {
            const int32_t grp = group_of_var[xindex.at(o.wrt)];
            if (grp >= 0) parts[grp].outputs.push_back(o);
          }
          for (size_t j = 0; j < G; ++j) {
            if (parts[j].outputs.empty()) continue;
            parts[j].root = partial[j];
            parts[j].scale_idx = row.scale_idx;
            r2.push_back(std::move(parts[j]));
          }
        }
        rows.swap(r2);
      }
    }
  }

  lap("V layout, separable sums");

  // ---- H_c: the Hessian of -y^T c_e - z^T c_i (problem.hpp:547-548, hessian.hpp:49-57) ----------------------------
  // The reference grows the gradient expressions of the WHOLE Lagrangian node by node (detail::gradient_tree,
  // variable_matrix.hpp:1757-1805) — and so did this function up to round 5: 0.8 million new nodes at cart-pole N=1000,
  // every stage's the same expressions on shifted variables, each then walked again for its row's pattern and sorted
  // into families by the tape compiler.  Now the families are found FIRST, on the model's own graph
  // (tape_families_analyze over the constraint values and the rows of A_e, A_i), and the symbolic reverse pass runs on
  // ONE member of every family (stage_gradient below: the same grad_expr rules, seeded with the member's multipliers);
  // its Hessian rows go through the flat compiler with the member, and the other members get theirs by position —
  // leaf bindings and destinations in V, nothing else: their gradient expressions are never built.  What is in no
  // family (boundary conditions, bounds, unique constraints) keeps the reference's way: the gradient tree of ITS part of
  // the Lagrangian.  SLPX_HESSIAN_FAMILIES=0: the whole Lagrangian that way, as before.
  // -y_ad^T c_e_ad - z_ad^T c_i_ad with the reference's matmul (variable_matrix.hpp:505-521: sum{0}; sum += lhs*rhs)
  // over the constraints `take` selects (null: all of them)
  auto lagrangian_rows = [&](const std::vector<uint8_t>* take_e, const std::vector<uint8_t>* take_i) {
    NodeId sum_e = g.constant(0.0);
    for (int j = 0; j < m_e; ++j)
      if (!take_e || (*take_e)[j]) sum_e = g.add(sum_e, g.mul(g.neg(s.y_nodes[j]), c_e[j]));
    NodeId sum_i = g.constant(0.0);
    for (int j = 0; j < m_i; ++j)
      if (!take_i || (*take_i)[j]) sum_i = g.add(sum_i, g.mul(s.z_nodes[j], c_i[j]));
    const NodeId lag = g.sub(sum_e, sum_i);
    return g.gradient_tree(g.topological_sort(lag), x);
  };
  // the hidden tail behind the H_c block, and every destination that was waiting for its place
  auto finish_layout = [&] {
    s.off_Hc = s.off_Hf + s.Hf.nnz();
    const int32_t off_tail = s.off_Hc + s.Hc.nnz();
    s.nV = off_tail + n_tail;
    s.V_static_raw.resize(s.nV, 0.0);
    s.V_scale_idx.resize(s.nV, -1);
    s.V_is_static.resize(s.nV, 1);
    for (int32_t j = 0; j < n_tail; ++j) {
      s.V_static_raw[off_tail + j] = tail_static[j];
      s.V_is_static[off_tail + j] = tail_is_static[j];
    }
    for (auto& v : live_vouts)
      if (v.dst >= kTailBase) v.dst = off_tail + (v.dst - kTailBase);
    for (auto& r : s.reduces)
      if (r.src_off >= kTailBase) r.src_off = off_tail + (r.src_off - kTailBase);
  };

  bool families_hessian = opt.families && m_e + m_i > 0;
  if (const char* env = std::getenv("SLPX_HESSIAN_FAMILIES")) families_hessian = families_hessian && env[0] != '0';
  if (const char* env = std::getenv("SLPX_TAPE_TEMPLATES")) families_hessian = families_hessian && env[0] != '0';
  TapeFamilySet fam_set;
  // per accepted family with Hessian rows: the positions (in a member's reachable set) of every row's variable and of
  // its outputs' variables, and of the multipliers (which value output of the member a multiplier leaf belongs to)
  struct FamilyHessian {
    std::vector<uint32_t> rows;                   // indices into `rows` (the representative's Hessian rows)
    std::vector<uint32_t> row_pos;                // per row: position of its variable
    std::vector<std::vector<uint32_t>> out_pos;   // per row, per output: position of the column's variable
    std::unordered_map<NodeId, uint32_t> multiplier_vout;  // y / z leaf of the representative -> position among its value outputs
  };
  std::vector<FamilyHessian> fam_hess;            // by family
  std::vector<uint32_t> remainder_rows;           // Hessian rows of what is in no family (indices into `rows`)
  bool used_families = false;
  if (families_hessian && live_vouts.size() + rows.size() >= 64 && tape_families_analyze(g, inputs, live_vouts, rows, fam_set)) {
    SetupLap lap_f;
    const size_t G0 = g.size();
    const size_t rows_at_analysis = rows.size();
    fam_hess.resize(fam_set.fams.size());
    std::vector<int32_t> xidx(G0, -1);  // node -> index of the decision variable
    for (int i = 0; i < n; ++i) xidx[x[i]] = i;
    // which constraint a value output is: (0 none, 1 equality, 2 inequality, index)
    auto constraint_of = [&](const TapeValueOut& v) -> std::pair<int, int> {
      if (v.scale_idx >= 1 && v.scale_idx <= m_e) return {1, v.scale_idx - 1};
      if (v.scale_idx > m_e && v.scale_idx <= m_e + m_i) return {2, v.scale_idx - 1 - m_e};
      return {0, 0};
    };
    std::vector<uint8_t> in_family_e(m_e, 0), in_family_i(m_i, 0);
    std::vector<int32_t> pos_of(G0, -1);
    std::vector<int32_t> walk_stamp;
    int32_t walk_stamp_next = 0;
    bool bail = false;
    for (size_t f = 0; f < fam_set.fams.size() && !bail; ++f) {
      const TapeFamilySet::Family& fam = fam_set.fams[f];
      if (fam.comps.size() < kTapeFamilyMin) continue;
      const uint32_t r = fam.rep;
      const NodeId* nodes = fam_set.all_nodes.data() + fam_set.all_start[r];
      const uint32_t cnt = fam_set.all_start[r + 1] - fam_set.all_start[r];
      FamilyHessian& fh = fam_hess[f];
      // ---- stage_gradient: the symbolic reverse pass (variable_matrix.hpp:1757-1805) on this member alone, in
      // descending node order (operands have smaller numbers: parents first), seeded with what the Lagrangian's own
      // reverse pass hands a constraint's root: -y_j, and (-1) z_j ----
      std::vector<NodeId> adj(cnt, kNull);
      for (uint32_t q = 0; q < cnt; ++q) pos_of[nodes[q]] = static_cast<int32_t>(q);
      bool any_seed = false;
      for (uint32_t q = fam_set.cvout_start[r]; q < fam_set.cvout_start[r + 1]; ++q) {
        const TapeValueOut& vo = live_vouts[fam_set.cvout[q]];
        const auto [kind, j] = constraint_of(vo);
        if (kind == 0 || g.type[vo.node] <= T_LINEAR) continue;  // (a LINEAR constraint has no second derivative)
        const NodeId mult = kind == 1 ? s.y_nodes[j] : s.z_nodes[j];
        const NodeId seed = kind == 1 ? g.neg(mult) : g.mul(g.constant(-1.0), mult);
        const int32_t pr = pos_of[vo.node];
        adj[pr] = g.add(adj[pr], seed);
        fh.multiplier_vout.emplace(mult, q - fam_set.cvout_start[r]);
        any_seed = true;
      }
      std::vector<uint32_t> extra;
      if (any_seed) {
        for (int32_t q = static_cast<int32_t>(cnt) - 1; q >= 0; --q) {
          const NodeId nd = nodes[q];
          const NodeId l = g.a0[nd], rr = g.a1[nd];
          if (l == kNull || adj[q] == kNull) continue;
          const NodeId gl = g.grad_expr(0, nd, adj[q]);
          adj[pos_of[l]] = g.add(adj[pos_of[l]], gl);
          if (rr != kNull) {
            const NodeId gr = g.grad_expr(1, nd, adj[q]);
            adj[pos_of[rr]] = g.add(adj[pos_of[rr]], gr);
          }
        }
        // the member's part of every Hessian row it reaches: the variables under adj(x_v)
        for (uint32_t q = 0; q < cnt && !bail; ++q) {
          const NodeId v = nodes[q];
          if (g.a0[v] != kNull || xidx[v] < 0 || adj[q] == kNull || g.type[adj[q]] == T_CONSTANT) continue;
          if (walk_stamp.size() < g.size()) walk_stamp.resize(g.size(), -1);
          const int32_t st = walk_stamp_next++;
          std::vector<NodeId> stack{adj[q]};
          walk_stamp[adj[q]] = st;
          std::vector<uint32_t> cols;
          while (!stack.empty()) {
            const NodeId w = stack.back();
            stack.pop_back();
            if (g.a0[w] == kNull) {
              if (static_cast<size_t>(w) < G0 && xidx[w] >= 0) {
                if (pos_of[w] < 0) bail = true;  // (a variable that is not of this member: not expected)
                else cols.push_back(static_cast<uint32_t>(pos_of[w]));
              }
              continue;
            }
            for (NodeId a : {g.a0[w], g.a1[w]})
              if (a != kNull && walk_stamp[a] != st) {
                walk_stamp[a] = st;
                stack.push_back(a);
              }
          }
          if (cols.empty()) continue;
          if (g.type[adj[q]] <= T_LINEAR) bail = true;  // (a row the reference would cache: not expected of a multiplier-weighted term)
          std::sort(cols.begin(), cols.end());
          TapeRow row;
          row.root = adj[q];
          row.scale_idx = -1;
          std::vector<uint32_t> outs;
          for (uint32_t pc : cols) {
            if (xidx[nodes[pc]] > xidx[v]) continue;  // lower triangle
            row.outputs.push_back({nodes[pc], 0});
            outs.push_back(pc);
          }
          if (row.outputs.empty()) continue;
          extra.push_back(static_cast<uint32_t>(rows.size()));
          fh.rows.push_back(static_cast<uint32_t>(rows.size()));
          fh.row_pos.push_back(q);
          fh.out_pos.push_back(std::move(outs));
          rows.push_back(std::move(row));
        }
      }
      for (uint32_t q = 0; q < cnt; ++q) pos_of[nodes[q]] = -1;
      if (bail) break;
      if (!tape_families_accept(g, inputs, live_vouts, rows, opt, fam_set, static_cast<uint32_t>(f), extra)) {
        fh = FamilyHessian{};  // (its constraints stay with the remainder; the rows built for it are never selected)
        continue;
      }
      for (uint32_t c : fam.comps)
        for (uint32_t q = fam_set.cvout_start[c]; q < fam_set.cvout_start[c + 1]; ++q) {
          const auto [kind, j] = constraint_of(live_vouts[fam_set.cvout[q]]);
          if (kind == 1) in_family_e[j] = 1;
          if (kind == 2) in_family_i[j] = 1;
        }
    }
    lap_f("  hessian families: representatives");
    if (!bail && !fam_set.accepted.empty()) {
      // ---- the remainder's part of the Lagrangian, the reference's way; LINEAR constraints left out (no second derivative)
      std::vector<uint8_t> take_e(m_e, 0), take_i(m_i, 0);
      bool any = false;
      for (int j = 0; j < m_e; ++j) any |= (take_e[j] = !in_family_e[j] && g.type[c_e[j]] > T_LINEAR);
      for (int j = 0; j < m_i; ++j) any |= (take_i[j] = !in_family_i[j] && g.type[c_i[j]] > T_LINEAR);
      MatrixBuild mHr;
      std::vector<NodeId> Hr_rows;
      if (any) {
        Hr_rows = lagrangian_rows(&take_e, &take_i);
        mHr = build_matrix(g, Hr_rows, x, n, n, true, visit);
      }
      lap_f("  hessian families: remainder");
      // ---- the pattern: the members' entries by position + the remainder's ----
      struct Ent {
        int32_t col, row;
      };
      std::vector<Ent> ents;
      for (const TapeFamilySet::Accepted& acc : fam_set.accepted) {
        const FamilyHessian& fh = fam_hess[acc.fam];
        if (fh.rows.empty()) continue;
        for (uint32_t c : fam_set.fams[acc.fam].comps) {
          const NodeId* nodes_c = fam_set.all_nodes.data() + fam_set.all_start[c];
          for (size_t k = 0; k < fh.rows.size() && !bail; ++k) {
            const int32_t xr = xidx[nodes_c[fh.row_pos[k]]];
            for (uint32_t pc : fh.out_pos[k]) {
              const int32_t xc = xidx[nodes_c[pc]];
              if (xr < 0 || xc < 0 || xc > xr) bail = true;  // (a member whose variables are ordered otherwise: not a shifted copy)
              ents.push_back({xc, xr});
            }
          }
        }
      }
      const size_t n_family_ents = ents.size();
      for (const RowEntry& e : mHr.entries) ents.push_back({e.col, e.row});
      CscPattern pat;
      pat.rows = pat.cols = n;
      pat.colptr.assign(n + 1, 0);
      for (const Ent& e : ents) ++pat.colptr[e.col + 1];
      for (int c = 0; c < n; ++c) pat.colptr[c + 1] += pat.colptr[c];
      pat.rowidx.resize(ents.size());
      {
        std::vector<int32_t> next(pat.colptr.begin(), pat.colptr.end() - 1);
        for (const Ent& e : ents) pat.rowidx[next[e.col]++] = e.row;
        for (int c = 0; c < n && !bail; ++c) {
          std::sort(pat.rowidx.begin() + pat.colptr[c], pat.rowidx.begin() + pat.colptr[c + 1]);
          // two sources of one entry (a variable that enters two families non-linearly, or a family and the remainder):
          // their sum is not something a tape output can be — the reference's way for the whole Lagrangian then
          if (std::adjacent_find(pat.rowidx.begin() + pat.colptr[c], pat.rowidx.begin() + pat.colptr[c + 1]) != pat.rowidx.begin() + pat.colptr[c + 1])
            bail = true;
        }
      }
      if (!bail) {
        (void)n_family_ents;
        s.Hc = pat;
        finish_layout();
        auto slot = [&](int32_t row, int32_t col) {
          const auto b0 = s.Hc.rowidx.begin() + s.Hc.colptr[col], e0 = s.Hc.rowidx.begin() + s.Hc.colptr[col + 1];
          return static_cast<int32_t>(std::lower_bound(b0, e0, row) - s.Hc.rowidx.begin());
        };
        // the families' entries are written by their members' tasks
        for (const TapeFamilySet::Accepted& acc : fam_set.accepted) {
          const FamilyHessian& fh = fam_hess[acc.fam];
          for (uint32_t c : fam_set.fams[acc.fam].comps) {
            const NodeId* nodes_c = fam_set.all_nodes.data() + fam_set.all_start[c];
            for (size_t k = 0; k < fh.rows.size(); ++k)
              for (uint32_t pc : fh.out_pos[k]) s.V_is_static[s.off_Hc + slot(xidx[nodes_c[fh.row_pos[k]]], xidx[nodes_c[pc]])] = 0;
          }
          s.nonlinear_rows += static_cast<int>(fh.rows.size() * fam_set.fams[acc.fam].comps.size());
        }
        // the remainder's rows, as add_matrix lays a matrix down — with its entries' places in the merged pattern
        const size_t rows_before = rows.size();
        if (!mHr.entries.empty() || !mHr.nonlinear_rows.empty()) {
          std::vector<int32_t> slot_of(mHr.entries.size());
          for (size_t k = 0; k < mHr.entries.size(); ++k) slot_of[k] = slot(mHr.entries[k].row, mHr.entries[k].col);
          add_matrix(mHr, Hr_rows, s.off_Hc, [](int32_t) { return -1; }, &slot_of);
        }
        for (size_t k = rows_before; k < rows.size(); ++k) remainder_rows.push_back(static_cast<uint32_t>(k));
        used_families = true;
        lap_f("  hessian families: pattern, layout");
      }
    }
    if (!used_families) {
      // (the families did not carry the Hessian: back to the reference's way; the rows built for representatives stay
      // behind unselected — compile_tape below must not see them)
      if (std::getenv("SLPX_SETUP_TIMING")) std::fprintf(stderr, "slpx setup: the Hessian's families were given up\n");
      rows.resize(rows_at_analysis);
      remainder_rows.clear();
      fam_set = TapeFamilySet{};
    }
  }
  if (!used_families) {
    std::vector<NodeId> Hc_rows = lagrangian_rows(nullptr, nullptr);
    lap("gradient tree of the Lagrangian (H_c rows)");
    MatrixBuild mHc = build_matrix(g, Hc_rows, x, n, n, true, visit);
    lap("row lists + pattern (H_c)");
    s.Hc = mHc.pat;
    finish_layout();
    add_matrix(mHc, Hc_rows, s.off_Hc, [](int32_t) { return -1; });
  }
  s.graph_nodes_after = g.size();
  lap("H_c: rows, pattern, layout");

  // the two tapes only read the graph: the small one (values only) compiles on a second thread
  // (SLPX_SETUP_THREADS=1: one after the other on this thread — the phase times then add up)
  const auto policy = SetupPool::get().threads() > 1 ? std::launch::async : std::launch::deferred;
  // (with the families of the model's graph in hand the values tape needs no analysis of its own: the same components,
  // their rows left out — classes found with the rows are classes without them)
  std::unique_ptr<TapeFamilySet> values_set;
  if (used_families) {
    values_set = std::make_unique<TapeFamilySet>();
    TapeFamilySet& v = *values_set;
    v.graph_size = fam_set.graph_size;
    v.ncomp = fam_set.ncomp;
    v.comp_start = fam_set.comp_start;
    v.members = fam_set.members;
    v.cvout_start = fam_set.cvout_start;
    v.cvout = fam_set.cvout;
    v.loose_vouts = fam_set.loose_vouts;
    v.crow_start.assign(v.ncomp + 1, 0);
    v.all_nodes = fam_set.all_nodes;
    v.all_start = fam_set.all_start;
    v.fams = fam_set.fams;
    v.param_order = fam_set.param_order;
    v.n_inputs = fam_set.n_inputs;
    v.comp_in_family.assign(v.ncomp, 0);
  }
  auto values_job = std::async(policy, [&] {
    SetupPool::InlineScope leave_the_pool_to_the_full_tape;
    if (values_set) {
      const std::vector<TapeRow> no_rows;
      for (size_t f = 0; f < values_set->fams.size(); ++f)
        tape_families_accept(g, inputs, live_vouts, no_rows, opt, *values_set, static_cast<uint32_t>(f), {});
      if (!values_set->accepted.empty()) {
        TapeProgram prog = tape_families_emit(g, inputs, live_vouts, no_rows, opt, *values_set, TapeFamilyHooks{});
        for (auto& v : live_vouts) prog.n_outputs = std::max(prog.n_outputs, v.dst + 1);
        return prog;
      }
    }
    return compile_tape(g, inputs, live_vouts, {}, opt);
  });
  std::future<void> patterns_job;
  if (on_patterns) patterns_job = std::async(policy, [&] { on_patterns(s); });
  if (used_families) {
    // the members of a family translate what is not of their own component: a multiplier (the same value output's, by
    // position), a constant the reverse pass made (the same constant), and the place of a Hessian row's output
    std::vector<int32_t> xidx(g.size(), -1);
    for (int i = 0; i < n; ++i) xidx[x[i]] = i;
    TapeFamilyHooks hooks;
    hooks.outside_leaf = [&](uint32_t f, NodeId rep_leaf, uint32_t comp) -> NodeId {
      const FamilyHessian& fh = fam_hess[f];
      const auto it = fh.multiplier_vout.find(rep_leaf);
      if (it == fh.multiplier_vout.end()) {
        if (g.op[rep_leaf] != OP_CONST) throw std::runtime_error("slpx: a Hessian family's representative reads a leaf its members cannot translate");
        return rep_leaf;
      }
      const TapeValueOut& vo = live_vouts[fam_set.cvout[fam_set.cvout_start[comp] + it->second]];
      return vo.scale_idx <= m_e ? s.y_nodes[vo.scale_idx - 1] : s.z_nodes[vo.scale_idx - 1 - m_e];
    };
    hooks.extra_dst = [&](uint32_t f, uint32_t k, uint32_t out, uint32_t comp) -> int32_t {
      const FamilyHessian& fh = fam_hess[f];
      const NodeId* nodes_c = fam_set.all_nodes.data() + fam_set.all_start[comp];
      const int32_t row = xidx[nodes_c[fh.row_pos[k]]], col = xidx[nodes_c[fh.out_pos[k][out]]];
      const auto b0 = s.Hc.rowidx.begin() + s.Hc.colptr[col], e0 = s.Hc.rowidx.begin() + s.Hc.colptr[col + 1];
      return s.off_Hc + static_cast<int32_t>(std::lower_bound(b0, e0, row) - s.Hc.rowidx.begin());
    };
    s.full = tape_families_emit(g, inputs, live_vouts, rows, opt, fam_set, hooks, {}, remainder_rows);
    for (auto& v : live_vouts) s.full.n_outputs = std::max(s.full.n_outputs, v.dst + 1);
  } else {
    s.full = compile_tape(g, inputs, live_vouts, rows, opt);
  }
  lap("tape compile (full)");
  s.values = values_job.get();
  if (patterns_job.valid()) patterns_job.get();
  lap("tape compile (values): the wait");
  s.full.n_inputs = s.values.n_inputs = s.n_inputs();
  s.full.n_outputs = s.values.n_outputs = s.nV;
  return s;
}

}  // namespace slpx
