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
    static thread_local std::vector<int32_t> seen;
    static thread_local int32_t stamp = 0;
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
  for (size_t c = 0; c < wrt.size(); ++c) visit.col[wrt[c]] = -1;
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
  // problem.hpp:547-548: -y_adᵀ c_e_ad - z_adᵀ c_i_ad with the reference's matmul
  // (variable_matrix.hpp:505-521: sum{0}; sum += lhs*rhs)
  NodeId lag;
  {
    std::vector<NodeId> neg_y(m_e);
    for (int j = 0; j < m_e; ++j) neg_y[j] = g.neg(s.y_nodes[j]);
    NodeId sum_e = g.constant(0.0);
    for (int j = 0; j < m_e; ++j) sum_e = g.add(sum_e, g.mul(neg_y[j], c_e[j]));
    NodeId sum_i = g.constant(0.0);
    for (int j = 0; j < m_i; ++j) sum_i = g.add(sum_i, g.mul(s.z_nodes[j], c_i[j]));
    lag = g.sub(sum_e, sum_i);
  }
  std::vector<NodeId> Hc_rows = g.gradient_tree(g.topological_sort(lag), x);
  s.graph_nodes_after = g.size();
  lap("gradient trees (Hessian rows)");

  // (kept by the thread between models: build_matrix leaves `col` blank again)
  static thread_local RowVisit visit;
  SetupLap lap_m;
  MatrixBuild mg = build_matrix(g, {f}, x, 1, n, false, visit);        // problem.hpp:535
  lap_m("  rows: g");
  MatrixBuild mHf = build_matrix(g, Hf_rows, x, n, n, true, visit);
  lap_m("  rows: H_f");
  MatrixBuild mHc = build_matrix(g, Hc_rows, x, n, n, true, visit);
  lap_m("  rows: H_c");
  MatrixBuild mAe = build_matrix(g, c_e, x, m_e, n, false, visit);     // problem.hpp:555
  lap_m("  rows: A_e");
  MatrixBuild mAi = build_matrix(g, c_i, x, m_i, n, false, visit);     // problem.hpp:560
  lap_m("  rows: A_i");
  lap("row lists + patterns");

  s.g_pat = mg.pat;
  s.Ae = mAe.pat;
  s.Ai = mAi.pat;
  s.Hf = mHf.pat;
  s.Hc = mHc.pat;
  s.off_f = 0;
  s.off_ce = 1;
  s.off_ci = s.off_ce + m_e;
  s.off_g = s.off_ci + m_i;
  s.off_Ae = s.off_g + s.g_pat.nnz();
  s.off_Ai = s.off_Ae + s.Ae.nnz();
  s.off_Hf = s.off_Ai + s.Ai.nnz();
  s.off_Hc = s.off_Hf + s.Hf.nnz();
  s.nV = s.off_Hc + s.Hc.nnz();
  s.V_static_raw.assign(s.nV, 0.0);
  s.V_scale_idx.assign(s.nV, -1);
  s.V_is_static.assign(s.nV, 1);

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
  auto add_matrix = [&](MatrixBuild& mb, const std::vector<NodeId>& roots, int off,
                        const std::function<int32_t(int32_t)>& scale_of_row) {
    std::vector<TapeRow> mrows(roots.size());
    for (size_t k = 0; k < mb.entries.size(); ++k) {
      const RowEntry& e = mb.entries[k];
      const int32_t dst = off + static_cast<int32_t>(k);
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
  add_matrix(mHc, Hc_rows, s.off_Hc, [](int32_t) { return -1; });

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
        const int32_t off_part = s.nV;
        s.nV += static_cast<int>(G);
        s.V_static_raw.resize(s.nV, 0.0);
        s.V_scale_idx.resize(s.nV, -1);
        s.V_is_static.resize(s.nV, 0);
        s.reduces.push_back({s.off_f, 0, off_part, static_cast<int32_t>(G)});
        std::vector<TapeValueOut> v2;
        for (auto& v : live_vouts)
          if (v.node != f) v2.push_back(v);
        for (size_t j = 0; j < G; ++j) {
          if (g.type[partial[j]] == T_CONSTANT) {
            s.V_static_raw[off_part + j] = g.val[partial[j]];
            s.V_is_static[off_part + j] = 1;
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
  // the two tapes only read the graph: the small one (values only) compiles on a second thread
  // (SLPX_SETUP_THREADS=1: one after the other on this thread — the phase times then add up)
  const auto policy = SetupPool::get().threads() > 1 ? std::launch::async : std::launch::deferred;
  auto values_job = std::async(policy, [&] {
    SetupPool::InlineScope leave_the_pool_to_the_full_tape;
    return compile_tape(g, inputs, live_vouts, {}, opt);
  });
  std::future<void> patterns_job;
  if (on_patterns) patterns_job = std::async(policy, [&] { on_patterns(s); });
  s.full = compile_tape(g, inputs, live_vouts, rows, opt);
  lap("tape compile (full)");
  s.values = values_job.get();
  if (patterns_job.valid()) patterns_job.get();
  lap("tape compile (values): the wait");
  s.full.n_inputs = s.values.n_inputs = s.n_inputs();
  s.full.n_outputs = s.values.n_outputs = s.nV;
  return s;
}

}  // namespace slpx
