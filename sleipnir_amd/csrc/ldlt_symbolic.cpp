#include "ldlt_symbolic.hpp"

#include "setup_timing.hpp"
#include "setup_threads.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <unordered_map>

namespace slpx {

namespace {

// the graph of the matrix: neighbours of v = idx[ptr[v] .. ptr[v + 1]), ascending, no duplicates
struct Adj {
  std::vector<int32_t> ptr, idx;
  struct Range {
    const int32_t *b, *e;
    const int32_t* begin() const { return b; }
    const int32_t* end() const { return e; }
    size_t size() const { return static_cast<size_t>(e - b); }
  };
  Range operator[](int32_t v) const { return {idx.data() + ptr[v], idx.data() + ptr[v + 1]}; }
  size_t size() const { return ptr.size() - 1; }
};

// ---------------------------------------------------------------------------
// Ordering: nested dissection with (constrained) minimum-degree leaves
// ---------------------------------------------------------------------------
struct Orderer {
  const Adj& adj;
  const std::vector<uint8_t>& has_diag;
  const LdltOptions& opt;
  std::vector<int32_t> stamp, dist;
  // The two sides of a separator are dissected on two threads (down to a few levels: parallel_depth): they
  // share no node and no edge, every call marks its nodes with a stamp of its own, and what the sides do
  // share — the `avail` counts of the separator nodes above them, which both add to and nobody reads before
  // both are through — is updated atomically.  The order is the sequential one: left, right, separator.
  std::atomic<int32_t> cur_stamp{0};
  int32_t new_stamp() { return cur_stamp.fetch_add(1, std::memory_order_relaxed) + 1; }
  std::vector<int32_t> order;
  // Pairing state for nodes whose unregularized diagonal is structurally zero
  // (constraint rows, variables that appear only linearly): such a node z may only be
  // eliminated after an ORIGINAL neighbour v that no other zero-diagonal node has
  // already claimed — then its pivot is -a_zv^2/d_v.  Two zero-diagonal nodes claiming
  // the same v would make the leading block [[d,a,b],[a,0,0],[b,0,0]] singular whatever
  // the values.  With distinct claims the claimed pairs form paths whose every prefix
  // has a perfect matching, so every leading principal block is structurally
  // nonsingular.
  std::vector<uint8_t> gone, claimed;
  std::vector<int32_t> avail;  // # eliminated, unclaimed original neighbours
  std::atomic<bool> forced{false};  // some node had to be eliminated without a partner
  int parallel_depth = 0;
  size_t parallel_min_nodes = 1500;

  Orderer(const Adj& a, const std::vector<uint8_t>& hd, const LdltOptions& o)
      : adj(a), has_diag(hd), opt(o), stamp(a.size(), 0), dist(a.size(), 0),
        gone(a.size(), 0), claimed(a.size(), 0), avail(a.size(), 0) {
    unsigned t = SetupPool::get().threads();
    while (t > 1) {
      ++parallel_depth;
      t >>= 1;
    }
  }

  bool ready(int32_t v) const { return !opt.defer_constraints || has_diag[v] || avail[v] > 0; }

  void eliminate(int32_t v, std::vector<int32_t>& out) {
    if (opt.defer_constraints && !has_diag[v]) {
      // claim the partner that the fewest other waiting zero-diagonal nodes could use
      int32_t partner = -1, partner_demand = 0;
      for (int32_t w : adj[v]) {
        if (!gone[w] || claimed[w]) continue;
        int32_t demand = 0;
        for (int32_t u : adj[w])
          if (!gone[u] && u != v && !has_diag[u]) ++demand;
        if (partner < 0 || demand < partner_demand) {
          partner = w;
          partner_demand = demand;
        }
      }
      if (partner < 0) {
        forced.store(true, std::memory_order_relaxed);
      } else {
        claimed[partner] = 1;
        for (int32_t u : adj[partner]) __atomic_fetch_sub(&avail[u], 1, __ATOMIC_RELAXED);
      }
    }
    gone[v] = 1;
    out.push_back(v);
    for (int32_t u : adj[v]) __atomic_fetch_add(&avail[u], 1, __ATOMIC_RELAXED);
  }

  // Minimum degree on the subgraph induced by `nodes`; with defer_constraints a
  // zero-diagonal node is only eligible while it has an unclaimed eliminated partner.
  void min_degree(const std::vector<int32_t>& nodes, std::vector<int32_t>& out) {
    const int m = static_cast<int>(nodes.size());
    if (m <= 64) {
      // the usual case (a dissection leaf): neighbour sets as bit masks, membership through a stamp of the leaf's own
      const int32_t s = new_stamp();
      for (int i = 0; i < m; ++i) {
        stamp[nodes[i]] = s;
        dist[nodes[i]] = i;  // (position in the leaf: dist is free once a set is down to a leaf)
      }
      uint64_t a[64];
      for (int i = 0; i < m; ++i) {
        uint64_t bits = 0;
        for (int32_t w : adj[nodes[i]])
          if (stamp[w] == s) bits |= 1ull << dist[w];
        a[i] = bits;
      }
      uint64_t done = 0;
      for (int step = 0; step < m; ++step) {
        int best = -1, best_deg = 0;
        for (int pass = 0; pass < 2 && best < 0; ++pass)
          for (int i = 0; i < m; ++i) {
            if (((done >> i) & 1) || (pass == 0 && !ready(nodes[i]))) continue;
            const int deg = __builtin_popcountll(a[i]);
            if (best < 0 || deg < best_deg) {
              best = i;
              best_deg = deg;
            }
          }
        done |= 1ull << best;
        eliminate(nodes[best], out);
        const uint64_t nb = a[best];
        for (uint64_t rest = nb; rest != 0; rest &= rest - 1) {
          const int u = __builtin_ctzll(rest);
          a[u] = (a[u] | nb) & ~((1ull << u) | (1ull << best));
        }
        a[best] = 0;
      }
      return;
    }
    std::unordered_map<int32_t, int32_t> loc;
    loc.reserve(m * 2);
    for (int i = 0; i < m; ++i) loc[nodes[i]] = i;
    std::vector<std::vector<int32_t>> a(m);
    for (int i = 0; i < m; ++i)
      for (int32_t w : adj[nodes[i]]) {
        auto it = loc.find(w);
        if (it != loc.end()) a[i].push_back(it->second);
      }
    for (auto& v : a) std::sort(v.begin(), v.end());
    std::vector<uint8_t> done(m, 0);
    std::vector<int32_t> merged;
    for (int step = 0; step < m; ++step) {
      int best = -1;
      for (int pass = 0; pass < 2 && best < 0; ++pass)
        for (int i = 0; i < m; ++i) {
          if (done[i] || (pass == 0 && !ready(nodes[i]))) continue;
          if (best < 0 || a[i].size() < a[best].size()) best = i;
        }
      done[best] = 1;
      eliminate(nodes[best], out);
      for (int32_t u : a[best]) {
        merged.clear();
        std::set_union(a[u].begin(), a[u].end(), a[best].begin(), a[best].end(),
                       std::back_inserter(merged));
        merged.erase(std::remove_if(merged.begin(), merged.end(),
                                    [&](int32_t w) { return w == u || w == best; }),
                     merged.end());
        a[u].swap(merged);
      }
      a[best].clear();
    }
  }

  // Separator nodes: variables with a diagonal first, then zero-diagonal nodes as they
  // find partners.
  void order_separator(std::vector<int32_t> sep, std::vector<int32_t>& out) {
    std::sort(sep.begin(), sep.end());
    std::vector<uint8_t> done(sep.size(), 0);
    for (size_t step = 0; step < sep.size(); ++step) {
      int best = -1;
      for (int pass = 0; pass < 3 && best < 0; ++pass)
        for (size_t i = 0; i < sep.size() && best < 0; ++i) {
          if (done[i]) continue;
          const int32_t v = sep[i];
          if (pass == 0 ? (has_diag[v] != 0) : (pass == 1 ? ready(v) : true)) best = static_cast<int>(i);
        }
      done[best] = 1;
      eliminate(sep[best], out);
    }
  }

  // BFS inside the current subset (marked by stamp == s): the nodes in visiting order, and where each level starts
  struct Levels {
    std::vector<int32_t> nodes, ptr;  // level l = nodes[ptr[l] .. ptr[l + 1])
    size_t size() const { return ptr.size() - 1; }
    size_t count(size_t l) const { return static_cast<size_t>(ptr[l + 1] - ptr[l]); }
    const int32_t* begin(size_t l) const { return nodes.data() + ptr[l]; }
    const int32_t* end(size_t l) const { return nodes.data() + ptr[l + 1]; }
  };
  void bfs_levels(int32_t start, int32_t s, size_t expected, Levels& L) {
    L.nodes.clear();
    L.ptr.clear();
    L.nodes.reserve(expected);
    const int32_t visited = new_stamp();
    dist[start] = visited;
    L.nodes.push_back(start);
    L.ptr.push_back(0);
    size_t lo = 0;
    while (lo < L.nodes.size()) {
      const size_t hi = L.nodes.size();
      L.ptr.push_back(static_cast<int32_t>(hi));
      for (size_t q = lo; q < hi; ++q)
        for (int32_t w : adj[L.nodes[q]])
          if (stamp[w] == s && dist[w] != visited) {
            dist[w] = visited;
            L.nodes.push_back(w);
          }
      lo = hi;
    }
  }

  void dissect(std::vector<int32_t> nodes, std::vector<int32_t>& out, int depth = 0) {
    if (nodes.empty()) return;
    if (static_cast<int>(nodes.size()) <= opt.leaf_size) {
      min_degree(nodes, out);
      return;
    }
    const int32_t s = new_stamp();
    for (int32_t v : nodes) stamp[v] = s;
    // pseudo-peripheral start: two BFS sweeps
    Levels levels;
    bfs_levels(nodes[0], s, nodes.size(), levels);
    const size_t reached = levels.nodes.size();
    if (reached < nodes.size()) {
      // disconnected: order each component on its own
      std::vector<int32_t> comp, rest;
      const int32_t mark = new_stamp();
      for (int32_t v : levels.nodes) {
        dist[v] = mark;
        comp.push_back(v);
      }
      for (int32_t v : nodes)
        if (dist[v] != mark) rest.push_back(v);
      dissect(std::move(comp), out, depth);
      dissect(std::move(rest), out, depth);
      return;
    }
    for (int sweep = 0; sweep < 2; ++sweep) {
      const int32_t far = *levels.begin(levels.size() - 1);
      bfs_levels(far, s, nodes.size(), levels);
    }
    if (levels.size() < 3) {
      min_degree(nodes, out);
      return;
    }
    // separator level: balance the two sides
    size_t total = nodes.size(), acc = 0, best_m = 1;
    size_t best_gap = total;
    for (size_t m = 0; m + 1 < levels.size(); ++m) {
      if (m >= 1) {
        size_t left = acc, right = total - acc - levels.count(m);
        size_t gap = left > right ? left - right : right - left;
        if (gap < best_gap) {
          best_gap = gap;
          best_m = m;
        }
      }
      acc += levels.count(m);
    }
    // shrink the separator to the nodes that actually touch the next level
    const int32_t nextmark = new_stamp();
    for (const int32_t* v = levels.begin(best_m + 1); v != levels.end(best_m + 1); ++v) dist[*v] = nextmark;
    std::vector<int32_t> sep, left, right;
    left.reserve(static_cast<size_t>(levels.ptr[best_m + 1]));
    for (const int32_t* pv = levels.begin(best_m); pv != levels.end(best_m); ++pv) {
      const int32_t v = *pv;
      bool touches = false;
      for (int32_t w : adj[v])
        if (stamp[w] == s && dist[w] == nextmark) {
          touches = true;
          break;
        }
      (touches ? sep : left).push_back(v);
    }
    left.insert(left.end(), levels.nodes.begin(), levels.nodes.begin() + levels.ptr[best_m]);
    right.assign(levels.nodes.begin() + levels.ptr[best_m + 1], levels.nodes.end());
    std::sort(left.begin(), left.end());
    std::sort(right.begin(), right.end());
    if (std::getenv("SLPX_ND_DEBUG")) std::fprintf(stderr, "nd depth %d: %zu nodes, %zu levels, sep %zu left %zu right %zu\n", depth, nodes.size(), levels.size(), sep.size(), left.size(), right.size());
    if (depth < parallel_depth && std::min(left.size(), right.size()) >= parallel_min_nodes) {
      std::vector<int32_t> right_out;
      std::exception_ptr failed;
      std::thread other([&] {
        try {
          dissect(std::move(right), right_out, depth + 1);
        } catch (...) {
          failed = std::current_exception();
        }
      });
      try {
        dissect(std::move(left), out, depth + 1);
      } catch (...) {
        other.join();
        throw;
      }
      other.join();
      if (failed) std::rethrow_exception(failed);
      out.insert(out.end(), right_out.begin(), right_out.end());
    } else {
      dissect(std::move(left), out, depth);
      dissect(std::move(right), out, depth);
    }
    // separator last
    order_separator(std::move(sep), out);
  }
};

}  // namespace

static LdltPlan build_ldlt_plan_once(const CscPattern& lower, int n_dec, const LdltOptions& opt_in,
                                     const std::vector<int32_t>* user_perm,
                                     const std::vector<uint8_t>* diag_has_source) {
  LdltOptions opt = opt_in;  // (single_problem_task_rules may adjust it after the first task partition)
  LdltPlan P;
  const int n = lower.cols;
  P.n = n;
  P.n_dec = n_dec;
  std::vector<uint8_t> has_diag(n, 0);
  for (int i = 0; i < n; ++i)
    has_diag[i] = diag_has_source ? (*diag_has_source)[i] : static_cast<uint8_t>(i < n_dec);

  SetupLap lap;
  // ---- ordering -------------------------------------------------------------
  bool ordering_forced = false;
  if (user_perm != nullptr && !user_perm->empty()) {
    P.perm = *user_perm;
  } else {
    Adj adj;
    {
      // (the pattern holds each pair once, below the diagonal: no duplicates to remove)
      adj.ptr.assign(n + 1, 0);
      for (int c = 0; c < n; ++c)
        for (int p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
          const int r = lower.rowidx[p];
          if (r != c) {
            ++adj.ptr[r + 1];
            ++adj.ptr[c + 1];
          }
        }
      for (int v = 0; v < n; ++v) adj.ptr[v + 1] += adj.ptr[v];
      adj.idx.resize(adj.ptr[n]);
      std::vector<int32_t> next(adj.ptr.begin(), adj.ptr.end() - 1);
      for (int c = 0; c < n; ++c)
        for (int p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
          const int r = lower.rowidx[p];
          if (r != c) {
            adj.idx[next[r]++] = c;
            adj.idx[next[c]++] = r;
          }
        }
      bool dup = false;
      for (int v = 0; v < n; ++v) {
        int32_t* b = adj.idx.data() + adj.ptr[v];
        int32_t* e = adj.idx.data() + adj.ptr[v + 1];
        if (!std::is_sorted(b, e)) std::sort(b, e);
        dup = dup || std::adjacent_find(b, e) != e;
      }
      if (dup) {  // (a pattern that names an entry twice: compact the lists)
        std::vector<int32_t> ptr2(n + 1, 0), idx2;
        for (int v = 0; v < n; ++v) {
          int32_t* b = adj.idx.data() + adj.ptr[v];
          int32_t* e = std::unique(b, adj.idx.data() + adj.ptr[v + 1]);
          idx2.insert(idx2.end(), b, e);
          ptr2[v + 1] = static_cast<int32_t>(idx2.size());
        }
        adj.ptr.swap(ptr2);
        adj.idx.swap(idx2);
      }
    }
    lap("  ldlt:   adjacency");
    Orderer ord(adj, has_diag, opt);
    // Hubs (LdltOptions::hub_factor) defeat level-set dissection — everything is two steps from
    // everything — so they are taken out of the graph that is dissected and eliminated last.
    std::vector<int32_t> rest, hubs;
    {
      std::vector<size_t> degs(n);
      for (int v = 0; v < n; ++v) degs[v] = adj[v].size();
      std::vector<size_t> sorted = degs;
      std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
      const size_t median = n ? sorted[n / 2] : 0;
      const size_t hub_degree =
          std::max<size_t>(opt.hub_floor, static_cast<size_t>(opt.hub_factor * static_cast<double>(median)));
      for (int v = 0; v < n; ++v) (degs[v] > hub_degree ? hubs : rest).push_back(v);
    }
    ord.order.reserve(n);
    lap("  ldlt:   hubs");
    ord.dissect(std::move(rest), ord.order);
    lap("  ldlt:   dissect");
    if (!hubs.empty()) ord.order_separator(std::move(hubs), ord.order);
    P.perm = std::move(ord.order);
    ordering_forced = ord.forced.load();
  }
  if (static_cast<int>(P.perm.size()) != n) throw std::runtime_error("ldlt: bad permutation size");
  P.iperm.assign(n, -1);
  for (int k = 0; k < n; ++k) P.iperm[P.perm[k]] = k;
  for (int k = 0; k < n; ++k)
    if (P.iperm[k] < 0) throw std::runtime_error("ldlt: permutation is not a bijection");

  lap("  ldlt: ordering (nested dissection)");
  // ---- permuted A: per column, strictly-lower rows with their lhs source ------
  std::vector<std::vector<std::pair<int32_t, int32_t>>> Acol(n);  // (row, src)
  std::vector<int32_t> diag_src(n, -1);
  for (int c = 0; c < n; ++c)
    for (int p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
      int32_t i = P.iperm[lower.rowidx[p]], j = P.iperm[c];
      if (i == j) diag_src[j] = p;
      else Acol[std::min(i, j)].emplace_back(std::max(i, j), p);
    }
  for (auto& v : Acol) std::sort(v.begin(), v.end());

  // ---- symbolic factorization: pattern of L and the etree -----------------------
  std::vector<std::vector<int32_t>> Lcol(n);
  std::vector<std::vector<int32_t>> children(n);
  P.parent.assign(n, -1);
  {
    std::vector<int32_t> tmp, tmp2;
    for (int j = 0; j < n; ++j) {
      tmp.clear();
      for (auto& [r, src] : Acol[j]) tmp.push_back(r);
      for (int32_t c : children[j]) {
        tmp2.clear();
        // Lcol[c] \ {j}
        std::set_union(tmp.begin(), tmp.end(), Lcol[c].begin() + 1, Lcol[c].end(),
                       std::back_inserter(tmp2));
        tmp.swap(tmp2);
      }
      Lcol[j] = tmp;
      if (!tmp.empty()) {
        P.parent[j] = tmp[0];
        children[tmp[0]].push_back(j);
      }
    }
  }
  // Which diagonal entries receive an update in the EXACT structure of L (before supernodes are relaxed
  // below: an explicit zero L(j, k) = 0 "updates" d_j by nothing).  A (2,2)-block diagonal without a
  // source and without such an update is a structurally zero pivot of the unregularized matrix.
  std::vector<uint8_t> diag_updated_exact(n, 0);
  for (int k = 0; k < n; ++k)
    for (int32_t r : Lcol[k]) diag_updated_exact[r] = 1;
  lap("  ldlt: pattern of L, elimination tree");
  // ---- relaxed supernodes (LdltOptions::relax_zeros) -------------------------------------------------
  int relaxed_merges = 0;
  int64_t relaxed_zeros = 0;
  if (opt.supernodal && opt.relax_zeros > 0) {
    for (int round = 0; round < 8; ++round) {
      // the chains as they stand (exact structure), their levels
      std::vector<int32_t> sn(n, -1);
      std::vector<int32_t> chain_ptr{0}, chain_cols;  // chain c = chain_cols[chain_ptr[c] .. chain_ptr[c + 1])
      chain_cols.reserve(n);
      for (int j = 0; j < n; ++j) {
        if (sn[j] >= 0) continue;
        const int32_t id = static_cast<int32_t>(chain_ptr.size()) - 1;
        sn[j] = id;
        chain_cols.push_back(j);
        size_t len = 1;
        int32_t k = j;
        while (len < opt.max_supernode_width) {
          const int32_t p = P.parent[k];
          if (p < 0 || sn[p] >= 0 || Lcol[k].size() != Lcol[p].size() + 1 ||
              len + 1 + Lcol[p].size() + 1 > opt.max_front_rows)
            break;
          sn[p] = id;
          chain_cols.push_back(p);
          ++len;
          k = p;
        }
        chain_ptr.push_back(static_cast<int32_t>(chain_cols.size()));
      }
      struct ChainView {
        const int32_t *b, *e;
        size_t size() const { return static_cast<size_t>(e - b); }
        int32_t operator[](size_t i) const { return b[i]; }
        int32_t back() const { return e[-1]; }
        const int32_t* begin() const { return b; }
        const int32_t* end() const { return e; }
      };
      struct ChainList {
        const std::vector<int32_t>&ptr, &col;
        size_t size() const { return ptr.size() - 1; }
        ChainView operator[](size_t c) const { return {col.data() + ptr[c], col.data() + ptr[c + 1]}; }
      } cols{chain_ptr, chain_cols};
      std::vector<int32_t> lvl(cols.size(), 0);
      for (int j = 0; j < n; ++j)
        for (int32_t c : children[j])
          if (sn[c] != sn[j]) lvl[sn[j]] = std::max(lvl[sn[j]], lvl[sn[c]] + 1);
      std::vector<int32_t> by_level(cols.size());
      std::iota(by_level.begin(), by_level.end(), 0);
      std::stable_sort(by_level.begin(), by_level.end(), [&](int32_t a, int32_t b) { return lvl[a] < lvl[b]; });
      std::vector<uint8_t> used(cols.size(), 0);
      int merged = 0;
      for (int32_t s : by_level) {
        if (used[s]) continue;
        const ChainView ch = cols[s];
        const int32_t p = P.parent[ch.back()];
        if (p < 0) continue;
        const int32_t J = sn[p];
        if (used[J] || cols[J][0] != p || lvl[s] + 1 < lvl[J]) continue;  // joins at the head, deepest child only
        const ChainView cj = cols[J];
        const size_t below = cj.size() + Lcol[cj.back()].size();  // rows under the joining columns
        if (ch.size() + cj.size() > opt.max_supernode_width || ch.size() + below + 1 > opt.max_front_rows) continue;
        int64_t zeros = 0;
        for (size_t i = 0; i < ch.size(); ++i)
          zeros += static_cast<int64_t>(ch.size() - 1 - i + below) - static_cast<int64_t>(Lcol[ch[i]].size());
        if (zeros > opt.relax_zeros) continue;
        std::vector<int32_t> tail(cj.begin(), cj.end());
        tail.insert(tail.end(), Lcol[cj.back()].begin(), Lcol[cj.back()].end());
        for (size_t i = 0; i < ch.size(); ++i) {
          std::vector<int32_t> rows(ch.begin() + i + 1, ch.end());
          rows.insert(rows.end(), tail.begin(), tail.end());
          Lcol[ch[i]] = std::move(rows);
        }
        used[s] = used[J] = 1;
        ++merged;
        relaxed_zeros += zeros;
      }
      relaxed_merges += merged;
      if (merged == 0) break;
    }
  }
  if (std::getenv("SLPX_LDLT_VERBOSE") && opt.relax_zeros > 0)
    std::fprintf(stderr, "ldlt relaxed supernodes: %d merges, %lld explicit zeros\n", relaxed_merges,
                 static_cast<long long>(relaxed_zeros));
  P.Lp.assign(n + 1, 0);
  for (int j = 0; j < n; ++j) P.Lp[j + 1] = P.Lp[j] + static_cast<int32_t>(Lcol[j].size());
  P.nnzL = P.Lp[n];
  P.Li.reserve(P.nnzL);
  for (int j = 0; j < n; ++j) P.Li.insert(P.Li.end(), Lcol[j].begin(), Lcol[j].end());
  {
    std::vector<int32_t> h(n, 0);
    for (int j = 0; j < n; ++j) {
      for (int32_t c : children[j]) h[j] = std::max(h[j], h[c] + 1);
      P.etree_height = std::max(P.etree_height, h[j] + 1);
    }
  }

  lap("  ldlt: relaxed supernodes");
  // ---- tasks: subtrees that fit in LDS, grouped in rounds -------------------------
  std::vector<int32_t> task_of(n, -1);
  std::vector<int32_t> round_of(n, -1);
  struct TaskCols {
    std::vector<int32_t> cols;
    uint32_t weight = 0;
    int round = 0;
  };
  std::vector<TaskCols> tcols;
  for (int partition_pass = 0; partition_pass < 2; ++partition_pass) {
    const uint32_t cap = opt.task_entries;
    std::fill(round_of.begin(), round_of.end(), -1);
    tcols.clear();
    std::vector<uint32_t> w(n), W(n);
    for (int j = 0; j < n; ++j) {
      // LDS cost in "entry equivalents" (~32 B): the column's entries plus the
      // update pairs it generates (8 B each), wherever those end up being stored
      const uint32_t c = static_cast<uint32_t>(Lcol[j].size());
      // (+1 entry and +c pairs for the right-hand-side row, see "entries" below)
      w[j] = (c + 2) + (c * (c + 1) / 2 + c + 3) / 4;
      if (w[j] > cap)
        throw std::runtime_error("ldlt: a single column exceeds the LDS task budget "
                                 "(global-memory supernode path not built yet)");
    }
    int assigned = 0, round = 0;
    while (assigned < n) {
      for (int j = 0; j < n; ++j) {
        if (round_of[j] >= 0) continue;
        W[j] = w[j];
        for (int32_t c : children[j])
          if (round_of[c] < 0) W[j] = std::min<uint64_t>(uint64_t(W[j]) + W[c], 0xffffffffu);
      }
      // unit roots: unassigned j with W <= cap whose parent is assigned-later or too heavy
      TaskCols cur;
      cur.round = round;
      std::vector<int32_t> stack;
      std::vector<int32_t> newly;
      for (int j = n - 1; j >= 0; --j) {  // visit roots before descendants
        if (round_of[j] >= 0 || W[j] > cap) continue;
        int32_t p = P.parent[j];
        if (p >= 0 && round_of[p] < 0 && W[p] <= cap) continue;  // inside a bigger unit
        // collect the unassigned subtree of j
        std::vector<int32_t> unit;
        stack.push_back(j);
        while (!stack.empty()) {
          int32_t v = stack.back();
          stack.pop_back();
          unit.push_back(v);
          for (int32_t c : children[v])
            if (round_of[c] < 0) stack.push_back(c);
        }
        if (cur.weight + W[j] > cap && !cur.cols.empty()) {
          tcols.push_back(std::move(cur));
          cur = TaskCols{};
          cur.round = round;
        }
        cur.weight += W[j];
        cur.cols.insert(cur.cols.end(), unit.begin(), unit.end());
        newly.insert(newly.end(), unit.begin(), unit.end());
      }
      if (!cur.cols.empty()) tcols.push_back(std::move(cur));
      if (newly.empty()) throw std::runtime_error("ldlt: task partition made no progress");
      for (int32_t v : newly) round_of[v] = round;
      assigned += static_cast<int>(newly.size());
      ++round;
    }
    P.n_rounds = round;
    if (partition_pass == 0 && opt.single_problem_task_rules) {
      bool again = false;
      if (opt.task_entries == LdltOptions{}.task_entries && tcols.size() > 400) {
        opt.task_entries *= 2;
        again = true;
      }
      if (opt.chain_from_deepest_child && opt.chain_from_deepest_min_round == 0 && tcols.size() > 250)
        opt.chain_from_deepest_min_round = 1;  // (read by the supernode pass below: no new partition for it)
      if (again) continue;
    }
    break;
  }
  const int ntasks = static_cast<int>(tcols.size());
  // column levels inside a task and local ordering
  std::vector<int32_t> tlevel(n, 0), lcol(n, -1);
  for (int t = 0; t < ntasks; ++t)
    for (int32_t j : tcols[t].cols) task_of[j] = t;
  lap("  ldlt: tasks and rounds");
  // ---- supernodes: chains of the etree inside one task whose columns share their structure ----
  // (|struct(L_j)| = |struct(L_parent)| + 1 makes the two structures equal up to the parent
  // itself, since struct(L_j) \ {parent} is always contained in struct(L_parent).)  A column
  // may have any number of other children: they just have to sit in lower levels.
  std::vector<int32_t> sn_of(n, -1), sn_pos(n, 0);
  std::vector<std::vector<int32_t>> sn_cols;
  // A column joins its parent's chain only from the DEEPEST side (opt.chain_from_deepest_child): a chain is one
  // front, scheduled after every child subtree of all its columns — a child k whose sibling subtree is as deep
  // as its own would otherwise put its pivots on the critical path behind that sibling, where alone it runs a
  // level earlier, beside it (seen at the ends of a dissected transcription: two separators of the same
  // structure in one front of 9, 1.4 us of the round's 6).  From round chain_from_deepest_min_round up: in the
  // leaf tasks, whose levels are full of fronts anyway, more and smaller fronts only cost tables and arena
  // (cart-pole N=5000 no longer fit two workgroups per CU).  depth = columns on the longest path of the
  // elimination tree below and including a column, inside its task.
  std::vector<int32_t> col_depth(n, 1);
  for (int j = 0; j < n; ++j)
    for (int32_t c : children[j])
      if (task_of[c] == task_of[j]) col_depth[j] = std::max(col_depth[j], col_depth[c] + 1);
  auto joins_from_deepest_side = [&](int32_t k, int32_t p) {
    if (!opt.chain_from_deepest_child || round_of[p] < opt.chain_from_deepest_min_round) return true;
    for (int32_t c : children[p])
      if (c != k && task_of[c] == task_of[p] && col_depth[c] >= col_depth[k]) return false;
    return true;
  };
  for (int j = 0; j < n; ++j) {
    if (sn_of[j] >= 0 || sn_of[j] == -2) continue;
    std::vector<int32_t> chain{j};
    sn_of[j] = static_cast<int32_t>(sn_cols.size());
    int32_t k = j;
    // A chain longer than the widest supernode is cut into pieces of (nearly) EQUAL width — nine columns
    // as 5 + 4, not 8 + 1: the dense pass of a front costs more than linearly in its width
    // (profiles/r04_microbench_front.txt), and the one-column tail was a level of its own.
    uint32_t cap = opt.max_supernode_width;
    if (opt.supernodal && opt.balance_supernode_cuts) {
      uint32_t len = 1;
      for (int32_t q = j;;) {
        const int32_t p = P.parent[q];
        if (p < 0 || task_of[p] != task_of[j] || sn_of[p] >= 0 || Lcol[q].size() != Lcol[p].size() + 1 ||
            len + 1 + Lcol[p].size() + 1 > opt.max_front_rows || !joins_from_deepest_side(q, p))
          break;
        ++len;
        q = p;
      }
      const uint32_t pieces = (len + opt.max_supernode_width - 1) / opt.max_supernode_width;
      cap = (len + pieces - 1) / pieces;
    }
    while (opt.supernodal && chain.size() < cap) {
      const int32_t p = P.parent[k];
      // rows of the trapezoid after adding p: chain + struct(L_p) + rhs row
      if (p < 0 || task_of[p] != task_of[j] || sn_of[p] >= 0 || Lcol[k].size() != Lcol[p].size() + 1 ||
          chain.size() + 1 + Lcol[p].size() + 1 > opt.max_front_rows || !joins_from_deepest_side(k, p))
        break;
      sn_of[p] = sn_of[j];
      sn_pos[p] = static_cast<int32_t>(chain.size());
      chain.push_back(p);
      k = p;
    }
    if (chain.size() > 1 && static_cast<int>(chain.size()) < opt.min_supernode_width) {
      // not worth a dense pass: back to single columns
      for (size_t i = 1; i < chain.size(); ++i) {
        sn_of[chain[i]] = -2;  // claimed by nobody; gets its own (lone) supernode below
        sn_pos[chain[i]] = 0;
      }
      chain.resize(1);
    }
    sn_cols.push_back(std::move(chain));
  }
  for (int j = 0; j < n; ++j)
    if (sn_of[j] == -2) {
      sn_of[j] = static_cast<int32_t>(sn_cols.size());
      sn_cols.push_back({j});
    }
  P.n_supernodes = static_cast<int>(sn_cols.size());
  for (auto& c : sn_cols) {
    P.widest_supernode = std::max<int>(P.widest_supernode, static_cast<int>(c.size()));
    if (P.sn_width_hist.size() <= c.size()) P.sn_width_hist.resize(c.size() + 1, 0);
    ++P.sn_width_hist[c.size()];
  }
  auto sn_width = [&](int32_t j) { return static_cast<int32_t>(sn_cols[sn_of[j]].size()); };
  // level of a supernode = 1 + the deepest supernode among the other children of its columns
  // (columns ascending: a child outside the chain ends its own chain, so that one is final)
  {
    std::vector<int32_t> snlevel(sn_cols.size(), 0);
    for (int j = 0; j < n; ++j)
      for (int32_t c : children[j])
        if (task_of[c] == task_of[j] && sn_of[c] != sn_of[j])
          snlevel[sn_of[j]] = std::max(snlevel[sn_of[j]], snlevel[sn_of[c]] + 1);
    for (int j = 0; j < n; ++j) tlevel[j] = snlevel[sn_of[j]];
  }
  for (int t = 0; t < ntasks; ++t) {
    auto& cols = tcols[t].cols;
    // within a level: supernodes by width (the lanes finishing them run width-specific code),
    // a supernode's columns consecutive and ascending
    std::sort(cols.begin(), cols.end(), [&](int32_t a, int32_t b) {
      if (tlevel[a] != tlevel[b]) return tlevel[a] < tlevel[b];
      if (sn_of[a] != sn_of[b]) {
        const int32_t wa = sn_width(a), wb = sn_width(b);
        return wa != wb ? wa < wb : sn_cols[sn_of[a]][0] < sn_cols[sn_of[b]][0];
      }
      return a < b;
    });
    for (size_t i = 0; i < cols.size(); ++i) lcol[cols[i]] = static_cast<int32_t>(i);
  }

  lap("  ldlt: supernodes, levels");
  // ---- entries ---------------------------------------------------------------------
  // local entry index of the diagonal of column j and of each L position.
  //
  // The right-hand side rides along as one extra ROW of the matrix: factorizing
  // [[K, b], [bᵀ, ·]] yields, as the last row of L, exactly z = D⁻¹L⁻¹Pb — the result of
  // the forward substitution and the diagonal scaling.  Column j therefore gets one more
  // entry U_b(j) = b_j − Σ_k U(j,k)·U_b(k)/d_k, computed at column j's level with the same
  // pair/contribution machinery as every other entry, and the forward-solve launches
  // disappear from the Newton step (they remain for re-solves with a new rhs).
  std::vector<uint32_t> diag_ent(n), lent(P.nnzL), bent(n);
  std::vector<uint32_t> task_nent(ntasks, 0);
  for (int t = 0; t < ntasks; ++t) {
    uint32_t e = 0;
    for (int32_t j : tcols[t].cols) {
      diag_ent[j] = e++;
      for (int32_t p = P.Lp[j]; p < P.Lp[j + 1]; ++p) lent[p] = e++;
      bent[j] = e++;
    }
    task_nent[t] = e;
    if (e > 65535u) throw std::runtime_error("ldlt: task exceeds 16-bit local indexing");
  }
  // Pair lists per entry, pseudo entries (update blocks for other tasks) per task, update slots per receiving
  // entry — as flat arrays (nested vectors of vectors were most of this pass: a million small allocations at
  // N=5000).  Entries are numbered globally, task by task: ge = task_ent_base[t] + local entry.
  std::vector<uint32_t> task_ent_base(ntasks + 1, 0);
  for (int t = 0; t < ntasks; ++t) task_ent_base[t + 1] = task_ent_base[t] + task_nent[t];
  const uint32_t total_ent = task_ent_base[ntasks];
  // Every pair is found by the task of its column k, the columns of a task in ascending order — the order the
  // single pass over all columns found them in, restricted to the task: the tasks in chunks on the setup threads.
  // A pseudo entry = (source task, receiving entry), numbered in the order of their first pair OVER ALL COLUMNS — also
  // their slot in the contribution buffer: the tasks note the column that made each of theirs, the numbers are handed
  // out afterwards by one walk over the columns.
  struct SlotTable {  // open addressing, key = receiving entry (global)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    size_t used = 0, mask = 0;
    SlotTable() { grow(1u << 10); }
    void clear() {
      std::fill(keys.begin(), keys.end(), ~0ull);
      used = 0;
    }
    void grow(size_t cap) {
      std::vector<uint64_t> ok;
      std::vector<uint32_t> ov;
      ok.swap(keys);
      ov.swap(vals);
      keys.assign(cap, ~0ull);
      vals.assign(cap, 0);
      mask = cap - 1;
      for (size_t i = 0; i < ok.size(); ++i)
        if (ok[i] != ~0ull) *find(ok[i]).second = ov[i];
    }
    static uint64_t mix(uint64_t k) {
      k ^= k >> 33;
      k *= 0xff51afd7ed558ccdull;
      k ^= k >> 33;
      return k;
    }
    // (is new, where its value lives)
    std::pair<bool, uint32_t*> find(uint64_t key) {
      size_t i = mix(key) & mask;
      while (keys[i] != ~0ull && keys[i] != key) i = (i + 1) & mask;
      const bool fresh = keys[i] == ~0ull;
      keys[i] = key;
      return {fresh, &vals[i]};
    }
    uint32_t* insert(uint64_t key, bool& fresh) {
      if (2 * (used + 1) > keys.size()) grow(2 * keys.size());
      auto r = find(key);
      fresh = r.first;
      used += fresh;
      return r.second;
    }
  };
  struct TaskPairs {
    std::vector<std::pair<uint32_t, LdltPair>> own, ext;  // (local entry | local pseudo entry, pair) as found
    std::vector<uint32_t> ext_target;                     // per local pseudo entry: receiving entry (global)
    std::vector<int32_t> ext_made_at;                     // ... and the column whose pair made it
    std::vector<uint32_t> ext_slot;                       // ... and its number (filled below)
  };
  std::vector<TaskPairs> tp(ntasks);
  std::vector<int32_t> cols_ptr(ntasks + 1, 0), cols_asc(n);  // the columns of a task, ascending
  for (int k = 0; k < n; ++k) ++cols_ptr[task_of[k] + 1];
  for (int t = 0; t < ntasks; ++t) cols_ptr[t + 1] += cols_ptr[t];
  {
    std::vector<int32_t> next(cols_ptr.begin(), cols_ptr.end() - 1);
    for (int k = 0; k < n; ++k) cols_asc[next[task_of[k]]++] = k;
  }
  std::atomic<bool> inconsistent{false};
  parallel_chunks(static_cast<size_t>(ntasks), 4, [&](size_t t_begin, size_t t_end, unsigned) {
    SlotTable slots;
    for (size_t tt = t_begin; tt < t_end; ++tt) {
      const int tk = static_cast<int>(tt);
      TaskPairs& T = tp[tt];
      slots.clear();
      for (int32_t ck = cols_ptr[tt]; ck < cols_ptr[tt + 1]; ++ck) {
        const int k = cols_asc[ck];
        const int32_t* rows = P.Li.data() + P.Lp[k];
        const int c = P.Lp[k + 1] - P.Lp[k];
        for (int a = 0; a < c; ++a) {
          const int32_t j = rows[a];  // target column
          const int tj = task_of[j];
          const uint32_t ent_jk = lent[P.Lp[k] + a];
          // updates between two columns of one supernode are done in registers (ldlt_kernels.h)
          if (sn_of[k] == sn_of[j]) continue;
          const uint32_t base_j = task_ent_base[tj];
          auto add_pair = [&](uint32_t target, const LdltPair& pr) {
            if (tk == tj) {
              T.own.emplace_back(target, pr);
            } else {
              bool fresh;
              uint32_t* slot = slots.insert(base_j + target, fresh);
              if (fresh) {
                *slot = static_cast<uint32_t>(T.ext_target.size());
                T.ext_target.push_back(base_j + target);
                T.ext_made_at.push_back(k);
              }
              T.ext.emplace_back(*slot, pr);
            }
          };
          // rhs row: U_b(j) -= U_b(k) · U(j,k) / d_k
          add_pair(bent[j], LdltPair{static_cast<uint16_t>(bent[k]), static_cast<uint16_t>(ent_jk),
                                     static_cast<uint16_t>(lcol[k]), 0});
          int32_t q = P.Lp[j];  // walk column j's rows
          for (int b = a; b < c; ++b) {
            const int32_t i = rows[b];
            uint32_t target;
            if (b == a) {
              target = diag_ent[j];
            } else {
              while (q < P.Lp[j + 1] && P.Li[q] != i) ++q;
              if (q >= P.Lp[j + 1]) {
                inconsistent = true;
                return;
              }
              target = lent[q];
            }
            add_pair(target, LdltPair{static_cast<uint16_t>(lent[P.Lp[k] + b]),
                                      static_cast<uint16_t>(ent_jk), static_cast<uint16_t>(lcol[k]), 0});
          }
        }
      }
    }
  });
  if (inconsistent) throw std::runtime_error("ldlt: fill pattern inconsistency");
  // the pseudo entries' numbers: in the order of the columns that made them (within a column: the order its task
  // made them in)
  std::vector<uint32_t> ext_task, ext_target;  // by slot: source task, receiving entry (global)
  {
    std::vector<uint32_t> cursor(ntasks, 0);
    for (int t = 0; t < ntasks; ++t) tp[t].ext_slot.resize(tp[t].ext_target.size());
    for (int k = 0; k < n; ++k) {
      const int t = task_of[k];
      TaskPairs& T = tp[t];
      uint32_t& cur = cursor[t];
      while (cur < T.ext_made_at.size() && T.ext_made_at[cur] == k) {
        T.ext_slot[cur] = P.n_contrib++;
        ext_task.push_back(static_cast<uint32_t>(t));
        ext_target.push_back(T.ext_target[cur]);
        ++cur;
      }
    }
  }
  // stable counting sorts: the pairs of an entry / of a pseudo entry in the order they were found — entries of a
  // task are consecutive, the pairs of a pseudo entry all its task's: every task writes its own ranges
  std::vector<uint32_t> own_ptr(total_ent + 1, 0), extp_ptr(P.n_contrib + 1, 0);
  std::vector<uint64_t> own_base(ntasks + 1, 0);
  for (int t = 0; t < ntasks; ++t) own_base[t + 1] = own_base[t] + tp[t].own.size();
  for (int t = 0; t < ntasks; ++t) {
    const TaskPairs& T = tp[t];
    std::vector<uint32_t> cnt(T.ext_target.size(), 0);
    for (auto& f : T.ext) ++cnt[f.first];
    for (size_t x = 0; x < cnt.size(); ++x) extp_ptr[T.ext_slot[x] + 1] = cnt[x];
  }
  for (uint32_t d = 0; d < P.n_contrib; ++d) extp_ptr[d + 1] += extp_ptr[d];
  std::vector<LdltPair> own_pairs(static_cast<size_t>(own_base[ntasks])), extp(extp_ptr[P.n_contrib]);
  parallel_chunks(static_cast<size_t>(ntasks), 4, [&](size_t t_begin, size_t t_end, unsigned) {
    std::vector<uint32_t> next;
    for (size_t tt = t_begin; tt < t_end; ++tt) {
      TaskPairs& T = tp[tt];
      const uint32_t base = task_ent_base[tt], ne = task_nent[tt];
      next.assign(ne + 1, 0);
      for (auto& f : T.own) ++next[f.first + 1];
      for (uint32_t e = 0; e < ne; ++e) next[e + 1] += next[e];
      const uint32_t first = static_cast<uint32_t>(own_base[tt]);
      for (uint32_t e = 0; e < ne; ++e) own_ptr[base + e] = first + next[e];
      for (auto& f : T.own) own_pairs[first + next[f.first]++] = f.second;
      next.assign(T.ext_target.size(), 0);
      for (size_t x = 0; x < next.size(); ++x) next[x] = extp_ptr[T.ext_slot[x]];
      for (auto& f : T.ext) extp[next[f.first]++] = f.second;
      T.own = {};
      T.ext = {};
    }
  });
  own_ptr[total_ent] = static_cast<uint32_t>(own_base[ntasks]);
  // the pseudo entries of a task (ascending slot = the order they were made in); the slots an entry takes
  std::vector<uint32_t> task_ext_ptr(ntasks + 1, 0), task_ext(P.n_contrib), ec_ptr(total_ent + 1, 0), ec(P.n_contrib);
  for (uint32_t d = 0; d < P.n_contrib; ++d) {
    ++task_ext_ptr[ext_task[d] + 1];
    ++ec_ptr[ext_target[d] + 1];
  }
  for (int t = 0; t < ntasks; ++t) task_ext_ptr[t + 1] += task_ext_ptr[t];
  for (uint32_t e = 0; e < total_ent; ++e) ec_ptr[e + 1] += ec_ptr[e];
  {
    std::vector<uint32_t> tn(task_ext_ptr.begin(), task_ext_ptr.end() - 1), en(ec_ptr.begin(), ec_ptr.end() - 1);
    for (uint32_t d = 0; d < P.n_contrib; ++d) {
      task_ext[tn[ext_task[d]]++] = d;
      ec[en[ext_target[d]]++] = d;
    }
  }
  tp = {};

  lap("  ldlt: entries, pair lists");
  // ---- solve lists ---------------------------------------------------------------
  // rows of L (CSR view): for row i the entries (i,k), k ascending
  std::vector<int32_t> Lrow_ptr(n + 1, 0);
  for (int32_t p = 0; p < P.nnzL; ++p) ++Lrow_ptr[P.Li[p] + 1];
  for (int i = 0; i < n; ++i) Lrow_ptr[i + 1] += Lrow_ptr[i];
  std::vector<std::pair<uint32_t, int32_t>> Lrow(static_cast<size_t>(P.nnzL));  // (lpos, k)
  {
    std::vector<int32_t> next(Lrow_ptr.begin(), Lrow_ptr.end() - 1);
    for (int k = 0; k < n; ++k)
      for (int32_t p = P.Lp[k]; p < P.Lp[k + 1]; ++p) Lrow[next[P.Li[p]]++] = {static_cast<uint32_t>(p), k};
  }
  // forward items of a column (same-task columns outside its supernode, then its chain), backward items (its
  // column of L), the pseudo rows a task computes for rows of other tasks: (source task, row), numbered in the
  // order they are first needed — their slot in the solve's contribution buffer
  std::vector<uint32_t> fwd_all_ptr(n + 1, 0), bwd_all_ptr(n + 1, 0);
  std::vector<LdltSolveItem> fwd_all, bwd_all;
  fwd_all.reserve(static_cast<size_t>(P.nnzL));
  bwd_all.reserve(static_cast<size_t>(P.nnzL));
  std::vector<std::pair<uint32_t, LdltSolveItem>> sext_found;
  std::vector<uint32_t> sext_task, sext_row;  // by slot
  SlotTable sslots;
  for (int j = 0; j < n; ++j) {
    const int tj = task_of[j];
    for (int32_t q = Lrow_ptr[j]; q < Lrow_ptr[j + 1]; ++q) {
      const uint32_t lpos = Lrow[q].first;
      const int32_t k = Lrow[q].second;
      const int tk = task_of[k];
      LdltSolveItem it{lpos, static_cast<uint32_t>(lcol[k])};
      if (tk == tj) {
        if (sn_of[k] != sn_of[j]) fwd_all.push_back(it);  // the chain's own columns follow below
      } else {
        bool fresh;
        uint32_t* slot = sslots.insert((static_cast<uint64_t>(tk) << 32) | static_cast<uint32_t>(j), fresh);
        if (fresh) {
          *slot = P.n_scontrib++;
          sext_task.push_back(static_cast<uint32_t>(tk));
          sext_row.push_back(static_cast<uint32_t>(j));
        }
        sext_found.emplace_back(*slot, it);
      }
    }
    // row j against the earlier columns of its own supernode: the LAST sn_pos[j] items, chain order
    for (int c = 0; c < sn_pos[j]; ++c) {
      const int32_t k = sn_cols[sn_of[j]][c];
      const int32_t* b = P.Li.data() + P.Lp[k];
      const int32_t* e = P.Li.data() + P.Lp[k + 1];
      const int32_t* f = std::lower_bound(b, e, j);
      if (f == e || *f != j) throw std::runtime_error("ldlt: supernode column lacks a chain row");
      fwd_all.push_back({static_cast<uint32_t>(f - P.Li.data()), static_cast<uint32_t>(lcol[k])});
    }
    fwd_all_ptr[j + 1] = static_cast<uint32_t>(fwd_all.size());
    for (int32_t p = P.Lp[j]; p < P.Lp[j + 1]; ++p) {
      int32_t i = P.Li[p];
      uint32_t ref = task_of[i] == tj ? static_cast<uint32_t>(lcol[i])
                                      : (0x80000000u | static_cast<uint32_t>(i));
      bwd_all.push_back({static_cast<uint32_t>(p), ref});
    }
    bwd_all_ptr[j + 1] = static_cast<uint32_t>(bwd_all.size());
  }
  std::vector<uint32_t> sext_item_ptr(P.n_scontrib + 1, 0);
  std::vector<LdltSolveItem> sext_item(sext_found.size());
  for (auto& f : sext_found) ++sext_item_ptr[f.first + 1];
  for (uint32_t d = 0; d < P.n_scontrib; ++d) sext_item_ptr[d + 1] += sext_item_ptr[d];
  {
    std::vector<uint32_t> next(sext_item_ptr.begin(), sext_item_ptr.end() - 1);
    for (auto& f : sext_found) sext_item[next[f.first]++] = f.second;
  }
  sext_found = {};
  std::vector<uint32_t> task_sext_ptr(ntasks + 1, 0), task_sext(P.n_scontrib), fc_ptr(n + 1, 0), fc(P.n_scontrib);
  for (uint32_t d = 0; d < P.n_scontrib; ++d) {
    ++task_sext_ptr[sext_task[d] + 1];
    ++fc_ptr[sext_row[d] + 1];
  }
  for (int t = 0; t < ntasks; ++t) task_sext_ptr[t + 1] += task_sext_ptr[t];
  for (int j = 0; j < n; ++j) fc_ptr[j + 1] += fc_ptr[j];
  {
    std::vector<uint32_t> tn(task_sext_ptr.begin(), task_sext_ptr.end() - 1), jn(fc_ptr.begin(), fc_ptr.end() - 1);
    for (uint32_t d = 0; d < P.n_scontrib; ++d) {
      task_sext[tn[sext_task[d]]++] = d;
      fc[jn[sext_row[d]]++] = d;
    }
  }

  size_t n_real_pairs = 0;
  lap("  ldlt: solve lists");
  // ---- flatten, tasks sorted by round ------------------------------------------------
  std::vector<int> torder(ntasks);
  std::iota(torder.begin(), torder.end(), 0);
  std::stable_sort(torder.begin(), torder.end(),
                   [&](int a, int b) { return tcols[a].round < tcols[b].round; });
  P.round_ptr.assign(P.n_rounds + 1, 0);
  for (int t = 0; t < ntasks; ++t) ++P.round_ptr[tcols[t].round + 1];
  for (int r = 0; r < P.n_rounds; ++r) P.round_ptr[r + 1] += P.round_ptr[r];

  // Every task's slices into a plan of its own (the same member names), the tasks in chunks on the setup threads;
  // joined below in round order with the alignment padding in front of every slice.
  std::vector<LdltPlan> flat(ntasks);
  parallel_chunks(static_cast<size_t>(ntasks), 4, [&](size_t ti_begin, size_t ti_end, unsigned) {
    for (size_t ti = ti_begin; ti < ti_end; ++ti) {
      const int t = torder[ti];
      LdltPlan& O = flat[ti];
      // 16-byte alignment of every per-task slice (the kernels stage them into LDS with
      // unrolled 16-byte loads)
      auto pad = [](auto& v, size_t multiple) {
        while (v.size() % multiple) v.push_back({});
      };
      while (O.ent_src.size() % 16) {
        O.ent_src.push_back(-1);
        O.ent_flags.push_back(0);
        O.ent_col.push_back(0);
        O.ent_out.push_back(0);
      }
      pad(O.pairs, 2);
      pad(O.ent_pair_ptr, 4);
      pad(O.ent_contrib_ptr, 4);
      while (O.lvl_ptr.size() % 4) {
        O.lvl_ptr.push_back(0);
        O.col_lvl_ptr.push_back(0);
      }
      pad(O.col_perm, 4);
      while (O.fwd_ptr.size() % 4) {
        O.fwd_ptr.push_back(0);
        O.fwd_contrib_ptr.push_back(0);
        O.bwd_ptr.push_back(0);
      }
      pad(O.fwd_items, 2);
      pad(O.bwd_items, 2);
      pad(O.sn_desc, 4);  // 12-byte records: four of them are three 16-byte groups
      pad(O.col_sn, 4);
      pad(O.sn_lvl_ptr, 4);
      LdltTask T{};
      const auto& cols = tcols[t].cols;
      T.round = static_cast<uint32_t>(tcols[t].round);
      T.n_col = static_cast<uint32_t>(cols.size());
      T.n_ent = task_nent[t];
      T.n_ext = task_ext_ptr[t + 1] - task_ext_ptr[t];
      T.ent_off = static_cast<uint32_t>(O.ent_src.size());
      T.col_off = static_cast<uint32_t>(O.col_perm.size());
      T.lvl_off = static_cast<uint32_t>(O.lvl_ptr.size());
      T.ext_off = static_cast<uint32_t>(O.ext_dst.size());
      T.pair_off = static_cast<uint32_t>(O.pairs.size());
      T.contrib_off = static_cast<uint32_t>(O.contrib_idx.size());
      T.fwd_item_off = static_cast<uint32_t>(O.fwd_items.size());
      T.bwd_item_off = static_cast<uint32_t>(O.bwd_items.size());
      T.sext_off = static_cast<uint32_t>(O.sext_dst.size());
      T.n_sext = task_sext_ptr[t + 1] - task_sext_ptr[t];
      T.sext_item_off = static_cast<uint32_t>(O.sext_items.size());
      T.scontrib_off = static_cast<uint32_t>(O.scontrib_idx.size());
      T.pair_ptr_off = static_cast<uint32_t>(O.ent_pair_ptr.size());
      T.contrib_ptr_off = static_cast<uint32_t>(O.ent_contrib_ptr.size());
      T.colptr_off = static_cast<uint32_t>(O.fwd_ptr.size());
      T.sext_ptr_off = static_cast<uint32_t>(O.sext_ptr.size());
      T.sn_off = static_cast<uint32_t>(O.sn_desc.size());
      if (O.col_sn.size() != O.col_perm.size() || O.sn_lvl_ptr.size() != O.lvl_ptr.size())
        throw std::runtime_error("ldlt: supernode arrays out of step with the column arrays");

      int32_t cur_level = -1;
      uint32_t pair_count = 0, contrib_count = 0, fwd_count = 0, bwd_count = 0, sc_count = 0;
      for (size_t ci = 0; ci < cols.size(); ++ci) {
        const int32_t j = cols[ci];
        if (tlevel[j] != cur_level) {
          O.lvl_ptr.push_back(diag_ent[j]);
          O.col_lvl_ptr.push_back(static_cast<uint32_t>(ci));
          O.sn_lvl_ptr.push_back(static_cast<uint32_t>(O.sn_desc.size()) - T.sn_off);
          cur_level = tlevel[j];
        }
        O.col_perm.push_back(static_cast<uint32_t>(j));
        const uint32_t w = static_cast<uint32_t>(sn_width(j)), pos = static_cast<uint32_t>(sn_pos[j]);
        O.col_sn.push_back(pos | (w << 8));
        if (w >= 2 && pos == 0) {
          // rows of the trapezoid: the chain, the common structure below it, the rhs row
          const uint32_t nr = w + static_cast<uint32_t>(Lcol[sn_cols[sn_of[j]].back()].size()) + 1;
          if (nr > kSnRowsMax) throw std::runtime_error("ldlt: supernode has more rows than a wave has lanes");
          O.sn_desc.push_back(LdltSn{diag_ent[j], static_cast<uint16_t>(w), static_cast<uint16_t>(nr),
                                     static_cast<uint16_t>(ci), 0});
        }
        auto emit_entry = [&](uint32_t le, int32_t src, uint8_t flags, uint32_t out) {
          O.ent_src.push_back(src);
          O.ent_flags.push_back(flags);
          O.ent_col.push_back(static_cast<uint16_t>(ci));
          O.ent_out.push_back(out);
          O.ent_pair_ptr.push_back(pair_count);
          O.ent_contrib_ptr.push_back(contrib_count);
          const uint32_t ge = task_ent_base[t] + le;
          O.pairs.insert(O.pairs.end(), own_pairs.begin() + own_ptr[ge], own_pairs.begin() + own_ptr[ge + 1]);
          pair_count += own_ptr[ge + 1] - own_ptr[ge];
          O.contrib_idx.insert(O.contrib_idx.end(), ec.begin() + ec_ptr[ge], ec.begin() + ec_ptr[ge + 1]);
          contrib_count += ec_ptr[ge + 1] - ec_ptr[ge];
        };
        const bool gamma_kind = P.perm[j] >= n_dec;
        // bit 3: entry of the diagonal block of a supernode with w >= 2 — finished (and written
        // out) by the lanes that work on the supernode, not by the generic passes
        const uint8_t in_block = w >= 2 ? 8 : 0;
        emit_entry(diag_ent[j], diag_src[j], static_cast<uint8_t>(1 | (gamma_kind ? 2 : 0) | in_block),
                   static_cast<uint32_t>(j));
        // off-diagonal entries: A source by merge with Acol[j]
        size_t ap = 0;
        for (int32_t p = P.Lp[j]; p < P.Lp[j + 1]; ++p) {
          int32_t src = -1;
          while (ap < Acol[j].size() && Acol[j][ap].first < P.Li[p]) ++ap;
          if (ap < Acol[j].size() && Acol[j][ap].first == P.Li[p]) src = Acol[j][ap].second;
          // the first w - pos - 1 rows of the column are the later columns of its chain
          const bool block_row = static_cast<uint32_t>(p - P.Lp[j]) + pos + 1 < w;
          emit_entry(lent[p], src, block_row ? in_block : 0, static_cast<uint32_t>(p));
        }
        // rhs-row entry: source = rhs[perm[j]], result z_j = U_b(j)/d_j
        emit_entry(bent[j], P.perm[j], 4, static_cast<uint32_t>(j));
        // solve lists
        O.fwd_ptr.push_back(fwd_count);
        O.fwd_items.insert(O.fwd_items.end(), fwd_all.begin() + fwd_all_ptr[j], fwd_all.begin() + fwd_all_ptr[j + 1]);
        fwd_count += fwd_all_ptr[j + 1] - fwd_all_ptr[j];
        O.fwd_contrib_ptr.push_back(sc_count);
        O.scontrib_idx.insert(O.scontrib_idx.end(), fc.begin() + fc_ptr[j], fc.begin() + fc_ptr[j + 1]);
        sc_count += fc_ptr[j + 1] - fc_ptr[j];
        O.bwd_ptr.push_back(bwd_count);
        O.bwd_items.insert(O.bwd_items.end(), bwd_all.begin() + bwd_all_ptr[j], bwd_all.begin() + bwd_all_ptr[j + 1]);
        bwd_count += bwd_all_ptr[j + 1] - bwd_all_ptr[j];
      }
      O.lvl_ptr.push_back(T.n_ent);
      O.col_lvl_ptr.push_back(T.n_col);
      T.n_lvl = static_cast<uint32_t>(O.lvl_ptr.size() - T.lvl_off - 1);
      T.n_sn = static_cast<uint32_t>(O.sn_desc.size()) - T.sn_off;
      O.sn_lvl_ptr.push_back(T.n_sn);
      // pseudo entries continue the pair_ptr array
      for (uint32_t q = task_ext_ptr[t]; q < task_ext_ptr[t + 1]; ++q) {
        const uint32_t d = task_ext[q];
        O.ext_dst.push_back(d);
        O.ent_pair_ptr.push_back(pair_count);
        O.pairs.insert(O.pairs.end(), extp.begin() + extp_ptr[d], extp.begin() + extp_ptr[d + 1]);
        pair_count += extp_ptr[d + 1] - extp_ptr[d];
      }
      O.ent_pair_ptr.push_back(pair_count);
      T.n_pairs = pair_count;
      T.n_fwd_items = fwd_count;
      T.n_bwd_items = bwd_count;
      O.ent_contrib_ptr.push_back(contrib_count);
      T.n_contrib_idx = contrib_count;
      while (O.contrib_idx.size() % 4 != 0) O.contrib_idx.push_back(0);  // 16-byte slices: staged into LDS
      O.fwd_ptr.push_back(fwd_count);
      O.fwd_contrib_ptr.push_back(sc_count);
      O.bwd_ptr.push_back(bwd_count);
      uint32_t sitem = 0;
      for (uint32_t q = task_sext_ptr[t]; q < task_sext_ptr[t + 1]; ++q) {
        const uint32_t d = task_sext[q];
        O.sext_dst.push_back(d);
        O.sext_ptr.push_back(sitem);
        O.sext_items.insert(O.sext_items.end(), sext_item.begin() + sext_item_ptr[d], sext_item.begin() + sext_item_ptr[d + 1]);
        sitem += sext_item_ptr[d + 1] - sext_item_ptr[d];
      }
      O.sext_ptr.push_back(sitem);
      O.max_lds_doubles = std::max(O.max_lds_doubles, T.n_ent + T.n_col);
      O.max_solve_lds_doubles = std::max(O.max_solve_lds_doubles, T.n_col);
      // LDS working sets of the staged kernels (see ldlt_kernels.h for the carve-up)
      auto q = [](uint32_t count, uint32_t per16) { return 16u * ((count + per16 - 1) / per16); };
      const uint32_t fb = q(pair_count, 2) + q(T.n_ent + T.n_ext + 1, 4) + q(T.n_lvl + 1, 4) +
                          q(T.n_ent, 4) + q(T.n_ent, 8) + q(T.n_ent, 16) + q(T.n_ent, 4) +
                          q(T.n_ent + 1, 4) + 8 * T.n_ent + 8 * T.n_col + 32 + q(3 * T.n_sn, 4) + q(T.n_lvl + 1, 4) +
                          q(T.n_contrib_idx, 4);
      const uint32_t items = std::max(fwd_count, bwd_count);
      const uint32_t sb = q(items, 2) + 2 * q(T.n_col + 1, 4) + q(T.n_lvl + 1, 4) + q(T.n_col, 4) +
                          8 * items + 8 * (T.n_col + 1) + 32 + q(T.n_col, 4) + q(3 * T.n_sn, 4) + q(T.n_lvl + 1, 4);
      O.factor_lds_bytes = std::max(O.factor_lds_bytes, fb);
      O.solve_lds_bytes = std::max(O.solve_lds_bytes, sb);
      O.tasks.push_back(T);
      O.flops = 2 * static_cast<int64_t>(pair_count);  // (carries the task's pair count to the join)
    }
  });
  // The join: where every task's slice of every array starts is known from the slices' lengths (16-byte alignment of
  // every slice: the kernels stage them into LDS with unrolled 16-byte loads), so the arrays are sized once and the
  // slices copied in, the tasks in chunks on the setup threads.
  {
    const size_t nt = flat.size();
    // offsets[a][ti], array a in the order of `place` below
    auto place = [&](auto member, size_t multiple, auto padval, std::vector<uint32_t>& offs) {
      offs.resize(nt);
      size_t off = 0;
      for (size_t ti = 0; ti < nt; ++ti) {
        off = (off + multiple - 1) / multiple * multiple;
        offs[ti] = static_cast<uint32_t>(off);
        off += (flat[ti].*member).size();
      }
      (P.*member).assign(off, padval);
    };
    std::vector<uint32_t> o_ent, o_flags, o_col, o_out, o_pairs, o_pptr, o_cptr, o_lvl, o_clvl, o_snlvl, o_colperm, o_colsn, o_sn,
        o_fptr, o_fcptr, o_bptr, o_fit, o_bit, o_ext, o_cidx, o_scidx, o_sdst, o_sptr, o_sit;
    place(&LdltPlan::ent_src, 16, int32_t{-1}, o_ent);
    place(&LdltPlan::ent_flags, 16, uint8_t{0}, o_flags);
    place(&LdltPlan::ent_col, 16, uint16_t{0}, o_col);
    place(&LdltPlan::ent_out, 16, uint32_t{0}, o_out);
    place(&LdltPlan::pairs, 2, LdltPair{}, o_pairs);
    place(&LdltPlan::ent_pair_ptr, 4, uint32_t{0}, o_pptr);
    place(&LdltPlan::ent_contrib_ptr, 4, uint32_t{0}, o_cptr);
    place(&LdltPlan::lvl_ptr, 4, uint32_t{0}, o_lvl);
    place(&LdltPlan::col_lvl_ptr, 4, uint32_t{0}, o_clvl);
    place(&LdltPlan::sn_lvl_ptr, 4, uint32_t{0}, o_snlvl);
    place(&LdltPlan::col_perm, 4, uint32_t{0}, o_colperm);
    place(&LdltPlan::col_sn, 4, uint32_t{0}, o_colsn);
    place(&LdltPlan::sn_desc, 4, LdltSn{}, o_sn);
    place(&LdltPlan::fwd_ptr, 4, uint32_t{0}, o_fptr);
    place(&LdltPlan::fwd_contrib_ptr, 4, uint32_t{0}, o_fcptr);
    place(&LdltPlan::bwd_ptr, 4, uint32_t{0}, o_bptr);
    place(&LdltPlan::fwd_items, 2, LdltSolveItem{}, o_fit);
    place(&LdltPlan::bwd_items, 2, LdltSolveItem{}, o_bit);
    place(&LdltPlan::ext_dst, 1, uint32_t{0}, o_ext);
    place(&LdltPlan::contrib_idx, 1, uint32_t{0}, o_cidx);
    place(&LdltPlan::scontrib_idx, 1, uint32_t{0}, o_scidx);
    place(&LdltPlan::sext_dst, 1, uint32_t{0}, o_sdst);
    place(&LdltPlan::sext_ptr, 1, uint32_t{0}, o_sptr);
    place(&LdltPlan::sext_items, 1, LdltSolveItem{}, o_sit);
    for (size_t ti = 0; ti < nt; ++ti)
      if (o_ent[ti] != o_flags[ti] || o_ent[ti] != o_col[ti] || o_ent[ti] != o_out[ti] || o_lvl[ti] != o_clvl[ti] ||
          o_lvl[ti] != o_snlvl[ti] || o_colperm[ti] != o_colsn[ti] || o_fptr[ti] != o_fcptr[ti] || o_fptr[ti] != o_bptr[ti])
        throw std::runtime_error("ldlt: supernode arrays out of step with the column arrays");
    P.tasks.resize(nt);
    parallel_chunks(nt, 4, [&](size_t ti_begin, size_t ti_end, unsigned) {
      for (size_t ti = ti_begin; ti < ti_end; ++ti) {
        LdltPlan& O = flat[ti];
        auto put = [&](auto member, uint32_t off) { std::copy((O.*member).begin(), (O.*member).end(), (P.*member).begin() + off); };
        put(&LdltPlan::ent_src, o_ent[ti]);
        put(&LdltPlan::ent_flags, o_flags[ti]);
        put(&LdltPlan::ent_col, o_col[ti]);
        put(&LdltPlan::ent_out, o_out[ti]);
        put(&LdltPlan::pairs, o_pairs[ti]);
        put(&LdltPlan::ent_pair_ptr, o_pptr[ti]);
        put(&LdltPlan::ent_contrib_ptr, o_cptr[ti]);
        put(&LdltPlan::lvl_ptr, o_lvl[ti]);
        put(&LdltPlan::col_lvl_ptr, o_clvl[ti]);
        put(&LdltPlan::sn_lvl_ptr, o_snlvl[ti]);
        put(&LdltPlan::col_perm, o_colperm[ti]);
        put(&LdltPlan::col_sn, o_colsn[ti]);
        put(&LdltPlan::sn_desc, o_sn[ti]);
        put(&LdltPlan::fwd_ptr, o_fptr[ti]);
        put(&LdltPlan::fwd_contrib_ptr, o_fcptr[ti]);
        put(&LdltPlan::bwd_ptr, o_bptr[ti]);
        put(&LdltPlan::fwd_items, o_fit[ti]);
        put(&LdltPlan::bwd_items, o_bit[ti]);
        put(&LdltPlan::ext_dst, o_ext[ti]);
        put(&LdltPlan::contrib_idx, o_cidx[ti]);
        put(&LdltPlan::scontrib_idx, o_scidx[ti]);
        put(&LdltPlan::sext_dst, o_sdst[ti]);
        put(&LdltPlan::sext_ptr, o_sptr[ti]);
        put(&LdltPlan::sext_items, o_sit[ti]);
        LdltTask T = O.tasks.at(0);
        T.ent_off += o_ent[ti];
        T.col_off += o_colperm[ti];
        T.lvl_off += o_lvl[ti];
        T.ext_off += o_ext[ti];
        T.pair_off += o_pairs[ti];
        T.contrib_off += o_cidx[ti];
        T.fwd_item_off += o_fit[ti];
        T.bwd_item_off += o_bit[ti];
        T.sext_off += o_sdst[ti];
        T.sext_item_off += o_sit[ti];
        T.scontrib_off += o_scidx[ti];
        T.pair_ptr_off += o_pptr[ti];
        T.contrib_ptr_off += o_cptr[ti];
        T.colptr_off += o_fptr[ti];
        T.sext_ptr_off += o_sptr[ti];
        T.sn_off += o_sn[ti];
        P.tasks[ti] = T;
      }
    });
    for (size_t ti = 0; ti < nt; ++ti) {
      const LdltPlan& O = flat[ti];
      P.max_lds_doubles = std::max(P.max_lds_doubles, O.max_lds_doubles);
      P.max_solve_lds_doubles = std::max(P.max_solve_lds_doubles, O.max_solve_lds_doubles);
      P.factor_lds_bytes = std::max(P.factor_lds_bytes, O.factor_lds_bytes);
      P.solve_lds_bytes = std::max(P.solve_lds_bytes, O.solve_lds_bytes);
      n_real_pairs += static_cast<size_t>(O.flops / 2);
    }
    flat = {};
  }

  lap("  ldlt: flattened plan arrays");
  // ---- multifrontal plan (LdltFront, ldlt_symbolic.hpp) -------------------------------------------
  // The fronts are an optional accelerator: whatever keeps them from being built — a limit (16-bit reach,
  // children per entry) or a structural corner case its consistency checks trip over (MfRefused) — leaves
  // P.mf = false and the pair-list plan above, which every system has, in charge.
  struct MfRefused {
    const char* why;
  };
  if (opt.multifrontal) try {
    struct Front {
      std::vector<int32_t> cols, R;  // permuted indices, ascending
      int task = 0, level = 0, parent = -1;
      uint32_t nr = 0, n_s = 0, s_base = 0;  // s_base: doubles into the arena
      std::vector<int> kids;
    };
    std::vector<Front> fronts;
    std::vector<int32_t> front_of(n, -1);
    std::vector<std::vector<int>> task_fronts(ntasks);  // level order
    bool ok = true;
    for (int t = 0; t < ntasks && ok; ++t)
      for (int32_t j : tcols[t].cols) {
        if (sn_pos[j] != 0) continue;
        Front f;
        f.cols = sn_cols[sn_of[j]];
        f.R = Lcol[f.cols.back()];
        f.task = t;
        f.level = tlevel[j];
        f.nr = static_cast<uint32_t>(f.cols.size() + f.R.size() + 1);
        const uint32_t r = static_cast<uint32_t>(f.R.size());
        f.n_s = r * (r + 1) / 2 + r;
        if (f.nr > kSnRowsMax || f.cols.size() > kSnWidthMax || f.n_s > 0xffffu) ok = false;
        for (int32_t c : f.cols) front_of[c] = static_cast<int>(fronts.size());
        task_fronts[t].push_back(static_cast<int>(fronts.size()));
        fronts.push_back(std::move(f));
      }
    for (size_t fi = 0; fi < fronts.size() && ok; ++fi) {
      Front& f = fronts[fi];
      const int32_t pc = P.parent[f.cols.back()];
      if (pc >= 0 && task_of[pc] == f.task && !f.R.empty()) {
        f.parent = front_of[pc];
        fronts[f.parent].kids.push_back(static_cast<int>(fi));
      }
    }
    // ---- levels of the fronts inside their task: 0 .. n_lvl - 1 in step with the column levels, then
    // balanced — a level is worked off sixteen fronts at a time (one wave each), and a front whose parent
    // sits more than one level up may as well run a level later: fronts with that slack leave levels of
    // more than sixteen (cart-pole N=1000: the leaf tasks' passes of sixteen 4..6 -> 4..5) — and inside a
    // level the wide fronts first, so that a second pass holds the cheap ones.
    for (int t = 0; t < ntasks && ok; ++t) {
      auto& fl = task_fronts[t];
      std::vector<int> distinct;
      for (int fi : fl) distinct.push_back(fronts[fi].level);
      std::sort(distinct.begin(), distinct.end());
      distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
      const int n_lvl = static_cast<int>(distinct.size());
      for (int fi : fl)
        fronts[fi].level = static_cast<int>(std::lower_bound(distinct.begin(), distinct.end(), fronts[fi].level) - distinct.begin());
      std::vector<int> count(n_lvl, 0);
      for (int fi : fl) ++count[fronts[fi].level];
      for (int l = 0; l + 1 < n_lvl; ++l) {
        if (count[l] <= 16) continue;
        // candidates: cheapest first (they add least to the level they join)
        std::vector<int> cand;
        for (int fi : fl) {
          const Front& f = fronts[fi];
          if (f.level != l) continue;
          const int limit = f.parent >= 0 ? fronts[f.parent].level : n_lvl;  // must stay below
          if (l + 1 < limit) cand.push_back(fi);
        }
        std::sort(cand.begin(), cand.end(), [&](int a, int b) {
          return fronts[a].cols.size() * 64 + fronts[a].R.size() < fronts[b].cols.size() * 64 + fronts[b].R.size();
        });
        for (int fi : cand) {
          if (count[l] <= 16 || count[l + 1] >= 16) break;
          ++fronts[fi].level;
          --count[l];
          ++count[l + 1];
        }
      }
      std::stable_sort(fl.begin(), fl.end(), [&](int a, int b) {
        const Front &fa = fronts[a], &fb = fronts[b];
        if (fa.level != fb.level) return fa.level < fb.level;
        return fa.cols.size() * 64 + fa.R.size() > fb.cols.size() * 64 + fb.R.size();
      });
    }
    // update slots between tasks: one per entry of a root front's block.  Every task builds its fronts' tables into
    // buffers of its own — the tasks in chunks on the setup threads — and the buffers are joined in task order
    // afterwards: offsets, slot numbers and the slots' receiving entries shifted by what came before.
    struct TaskOut {
      std::vector<uint32_t> mf_anc, mf_lvl_ptr, mf_ext;
      std::vector<uint16_t> mf_tab;
      std::vector<LdltFront> mf_fronts;
      std::vector<std::pair<uint32_t, uint32_t>> mc_found;  // (receiving entry, global; slot of this task) in slot order
      uint32_t n_slots = 0, n_mfma = 0, max_nch = 0, max_front_rows = 0;
      bool ok = true;
      const char* refused = nullptr;
    };
    struct Scratch {
      std::vector<uint16_t> cell_vals;  // per front: the children's values of every table cell, `kids` slots a cell
      std::vector<uint8_t> cell_cnt;
      std::vector<uint32_t> to;
    };
    auto entry_of = [&](int32_t row, int32_t col) -> uint32_t {  // row = -1: the right-hand-side row
      if (row == col) return diag_ent[col];
      if (row < 0) return bent[col];
      const int32_t* b = P.Li.data() + P.Lp[col];
      const int32_t* e = P.Li.data() + P.Lp[col + 1];
      const int32_t* f = std::lower_bound(b, e, row);
      if (f == e || *f != row) throw MfRefused{"a front's update block has an entry outside the pattern of L"};
      return lent[f - P.Li.data()];
    };
    P.mf_tasks.assign(ntasks, LdltMfTask{});
    std::vector<TaskOut> outs(P.tasks.size());
    auto build_task = [&](size_t ti, Scratch& S) {
      const int t = torder[ti];
      const LdltTask& T = P.tasks[ti];
      LdltMfTask& M = P.mf_tasks[ti];
      TaskOut& O = outs[ti];
      const auto& fl = task_fronts[t];
      // ---- S blocks: a block lives from its front's level to its parent's; first fit over the free gaps
      uint32_t arena = 2;  // [0] = 0.0, [1] = scratch
      {
        struct Blk { uint32_t off, len; int until; };
        std::vector<Blk> live;
        for (int fi : fl) {
          Front& f = fronts[fi];
          if (f.parent < 0 || f.n_s == 0) continue;  // root fronts write to the update slots
          const int lvl = f.level, until = fronts[f.parent].level;
          live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.until < lvl; }), live.end());
          std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
          uint32_t at = 2;
          for (const Blk& b : live) {
            if (at + f.n_s <= b.off) break;
            at = std::max(at, b.off + b.len);
          }
          f.s_base = at;
          live.push_back({at, f.n_s, until});
          arena = std::max(arena, at + f.n_s);
        }
      }
      M.arena = arena;
      // ---- rows of ancestor tasks the solve reaches
      std::vector<int32_t> anc;
      for (int fi : fl)
        for (int32_t i : fronts[fi].R)
          if (task_of[i] != t) anc.push_back(i);
      std::sort(anc.begin(), anc.end());
      anc.erase(std::unique(anc.begin(), anc.end()), anc.end());
      M.anc_off = static_cast<uint32_t>(O.mf_anc.size());
      M.n_anc = static_cast<uint32_t>(anc.size());
      for (int32_t i : anc) O.mf_anc.push_back(static_cast<uint32_t>(i));
      // ---- the 64 KB every table offset must reach
      const uint32_t off_arena = 8u * T.n_ent, off_invd = off_arena + 8u * arena, off_x = off_invd + 8u * T.n_col;
      const uint32_t reach = off_x + 8u * (T.n_col + M.n_anc + 1u);
      if (reach > 0x10000u) {
        O.ok = false;
        return;
      }
      const uint16_t kZero = static_cast<uint16_t>(off_arena), kScratch = static_cast<uint16_t>(off_arena + 8u);
      auto s_addr = [&](const Front& f, uint32_t e) { return static_cast<uint16_t>(off_arena + 8u * (f.s_base + e)); };
      auto u_addr = [&](uint32_t ent) { return static_cast<uint16_t>(8u * ent); };
      // ---- fronts and their tables
      M.front_off = static_cast<uint32_t>(O.mf_fronts.size());
      M.n_front = static_cast<uint32_t>(fl.size());
      M.tab_off = static_cast<uint32_t>(O.mf_tab.size());
      M.ext_off = static_cast<uint32_t>(O.mf_ext.size());
      int cur_level = -1;
      uint32_t n_ext = 0;
      for (size_t q = 0; q < fl.size(); ++q) {
        const Front& f = fronts[fl[q]];
        if (f.level != cur_level) {
          O.mf_lvl_ptr.push_back(static_cast<uint32_t>(q));
          cur_level = f.level;
        }
        const uint32_t w = static_cast<uint32_t>(f.cols.size()), r = static_cast<uint32_t>(f.R.size()), nr = f.nr;
        const uint32_t base0 = diag_ent[f.cols[0]];
        auto tr = [&](uint32_t c, uint32_t row) { return base0 + c * nr - (c * (c - 1)) / 2 + (row - c); };
        // where the rows of each child land in this front (a child's block has at most one value for a cell:
        // the cells' lists are the children's values in child order)
        const size_t n_kids = f.kids.size(), n_piv = static_cast<size_t>(nr) * w, n_cells = n_piv + f.n_s;
        S.cell_cnt.assign(n_cells, 0);
        if (S.cell_vals.size() < n_cells * n_kids) S.cell_vals.resize(n_cells * n_kids);
        for (int ki : f.kids) {
          const Front& k = fronts[ki];
          const uint32_t rk = static_cast<uint32_t>(k.R.size());
          std::vector<uint32_t>& to = S.to;
          to.resize(rk + 1);
          for (uint32_t a = 0; a < rk; ++a) {
            const int32_t i = k.R[a];
            auto ic = std::lower_bound(f.cols.begin(), f.cols.end(), i);
            if (ic != f.cols.end() && *ic == i) {
              to[a] = static_cast<uint32_t>(ic - f.cols.begin());
            } else {
              auto ir = std::lower_bound(f.R.begin(), f.R.end(), i);
              if (ir == f.R.end() || *ir != i) throw MfRefused{"a child front's row is not in its parent front"};
              to[a] = w + static_cast<uint32_t>(ir - f.R.begin());
            }
          }
          to[rk] = nr - 1;
          for (uint32_t a = 0; a <= rk; ++a)
            for (uint32_t b = 0; b < rk && b <= a; ++b) {
              const uint32_t ta = to[a], tb = to[b];
              const uint16_t src = s_addr(k, a * (a + 1) / 2 + b);
              const size_t cell = tb < w ? static_cast<size_t>(ta) * w + tb : n_piv + (ta - w) * (ta - w + 1) / 2 + (tb - w);
              if (S.cell_cnt[cell] == 255 || S.cell_cnt[cell] >= n_kids) {
                O.ok = false;
                break;
              }
              S.cell_vals[cell * n_kids + S.cell_cnt[cell]++] = src;
            }
        }
        if (!O.ok) break;
        uint32_t nch = 0;
        for (size_t c = 0; c < n_cells; ++c) nch = std::max<uint32_t>(nch, S.cell_cnt[c]);
        auto piv_at = [&](uint32_t row, uint32_t c, uint32_t kk) {
          const size_t cell = static_cast<size_t>(row) * w + c;
          return kk < S.cell_cnt[cell] ? S.cell_vals[cell * n_kids + kk] : kZero;
        };
        auto upd_at = [&](uint32_t e, uint32_t kk) {
          const size_t cell = n_piv + e;
          return kk < S.cell_cnt[cell] ? S.cell_vals[cell * n_kids + kk] : kZero;
        };
        LdltFront F{};
        F.tab = static_cast<uint32_t>(O.mf_tab.size()) - M.tab_off;
        F.base0 = static_cast<uint16_t>(base0);
        F.col0 = static_cast<uint16_t>(lcol[f.cols[0]]);
        F.w = static_cast<uint8_t>(w);
        F.nr = static_cast<uint8_t>(nr);
        F.nch = static_cast<uint8_t>(nch);
        F.n_s = static_cast<uint16_t>(f.n_s);
        const bool root = f.parent < 0;
        // the update block on the matrix cores: four pivot columns or more under more entries than two
        // lane-per-entry passes cover (ldlt_mf_kernels.h: mf_update_mfma)
        const bool mfma = w >= 4 && f.n_s > opt.mfma_min_entries;
        F.flags = static_cast<uint8_t>((root ? 1 : 0) | (mfma ? 2 : 0));
        O.n_mfma += mfma;
        F.ext = static_cast<uint16_t>(n_ext);
        if (root && n_ext + f.n_s > 0xffffu) {
          O.ok = false;
          break;
        }
        // pivot table: rows x [k][c]
        for (uint32_t row = 0; row < nr; ++row)
          for (uint32_t k = 0; k <= nch; ++k)
            for (uint32_t c = 0; c < w; ++c) {
              uint16_t v;
              if (k == 0) v = row >= c ? u_addr(tr(c, row)) : kScratch;
              else v = row >= c ? piv_at(row, c, k - 1) : kZero;
              O.mf_tab.push_back(v);
            }
        // update table: out, U(a, 0), U(b, 0), children
        for (uint32_t a = 0; a <= r; ++a)
          for (uint32_t b = 0; b < r && b <= a; ++b) {
            const uint32_t e = a * (a + 1) / 2 + b;
            O.mf_tab.push_back(root ? static_cast<uint16_t>(e) : s_addr(f, e));
            O.mf_tab.push_back(u_addr(tr(0, w + a)));
            O.mf_tab.push_back(u_addr(tr(0, w + b)));
            for (uint32_t k = 0; k < nch; ++k) O.mf_tab.push_back(upd_at(e, k));
            if (root) {
              const int32_t gb = f.R[b], ga = a < r ? f.R[a] : -1;
              const int tj = task_of[gb];
              const uint32_t slot = O.n_slots++;
              O.mc_found.emplace_back(task_ent_base[tj] + entry_of(ga, gb), slot);
              O.mf_ext.push_back(slot);
              ++n_ext;
            }
          }
        // solve table: x of the rows of R
        for (uint32_t a = 0; a < r; ++a) {
          const int32_t i = f.R[a];
          uint32_t xi;
          if (task_of[i] == t) xi = static_cast<uint32_t>(lcol[i]);
          else xi = T.n_col + static_cast<uint32_t>(std::lower_bound(anc.begin(), anc.end(), i) - anc.begin());
          O.mf_tab.push_back(static_cast<uint16_t>(off_x + 8u * xi));
        }
        O.mf_fronts.push_back(F);
        O.max_nch = std::max(O.max_nch, nch);
        O.max_front_rows = std::max(O.max_front_rows, nr);
      }
      if (!O.ok) return;
      O.mf_lvl_ptr.push_back(static_cast<uint32_t>(fl.size()));
      if (O.mf_lvl_ptr.size() != T.n_lvl + 1)
        throw MfRefused{"front levels out of step with the column levels"};
      M.n_ext = n_ext;
      M.n_tab = static_cast<uint32_t>(O.mf_tab.size()) - M.tab_off;
    };
    if (ok)
      parallel_chunks(P.tasks.size(), 4, [&](size_t t_begin, size_t t_end, unsigned) {
        Scratch S;
        for (size_t ti = t_begin; ti < t_end; ++ti) {
          try {
            build_task(ti, S);
          } catch (const MfRefused& r) {
            outs[ti].refused = r.why;
          }
        }
      });
    std::vector<std::pair<uint32_t, uint32_t>> mc_found;  // (receiving entry, global; slot) in slot order
    for (size_t ti = 0; ti < P.tasks.size() && ok; ++ti) {
      TaskOut& O = outs[ti];
      if (O.refused != nullptr) throw MfRefused{O.refused};
      if (!O.ok) {
        ok = false;
        break;
      }
      const LdltTask& T = P.tasks[ti];
      LdltMfTask& M = P.mf_tasks[ti];
      M.anc_off = static_cast<uint32_t>(P.mf_anc.size());
      P.mf_anc.insert(P.mf_anc.end(), O.mf_anc.begin(), O.mf_anc.end());
      while (P.mf_anc.size() % 4) P.mf_anc.push_back(0);
      M.front_off = static_cast<uint32_t>(P.mf_fronts.size());
      M.tab_off = static_cast<uint32_t>(P.mf_tab.size());
      M.ext_off = static_cast<uint32_t>(P.mf_ext.size());
      while (P.mf_lvl_ptr.size() < T.lvl_off) P.mf_lvl_ptr.push_back(0);
      P.mf_lvl_ptr.insert(P.mf_lvl_ptr.end(), O.mf_lvl_ptr.begin(), O.mf_lvl_ptr.end());
      if (P.mf_lvl_ptr.size() != T.lvl_off + T.n_lvl + 1) throw MfRefused{"front levels out of step with the column levels"};
      P.mf_fronts.insert(P.mf_fronts.end(), O.mf_fronts.begin(), O.mf_fronts.end());
      P.mf_tab.insert(P.mf_tab.end(), O.mf_tab.begin(), O.mf_tab.end());
      for (uint32_t slot : O.mf_ext) P.mf_ext.push_back(P.mf_n_contrib + slot);
      for (auto& f : O.mc_found) mc_found.emplace_back(f.first, P.mf_n_contrib + f.second);
      P.mf_n_contrib += O.n_slots;
      P.mf_n_mfma += O.n_mfma;
      P.mf_max_nch = std::max(P.mf_max_nch, O.max_nch);
      P.mf_max_front_rows = std::max(P.mf_max_front_rows, O.max_front_rows);
      while (P.mf_tab.size() % 8) P.mf_tab.push_back(0);
      while (P.mf_ext.size() % 4) P.mf_ext.push_back(0);
      O = TaskOut{};
    }
    // update slots per receiving entry (the tasks' entries are numbered in emission order)
    std::vector<uint32_t> mc_ptr(total_ent + 1, 0), mc(mc_found.size());
    for (auto& f : mc_found) ++mc_ptr[f.first + 1];
    for (uint32_t e = 0; e < total_ent; ++e) mc_ptr[e + 1] += mc_ptr[e];
    {
      std::vector<uint32_t> next(mc_ptr.begin(), mc_ptr.end() - 1);
      for (auto& f : mc_found) mc[next[f.first]++] = f.second;
    }
    for (size_t ti = 0; ti < P.tasks.size() && ok; ++ti) {
      const int t = torder[ti];
      LdltMfTask& M = P.mf_tasks[ti];
      while (P.mf_contrib_ptr.size() % 4) P.mf_contrib_ptr.push_back(0);
      while (P.mf_cent.size() % 8) P.mf_cent.push_back(0);
      M.contrib_ptr_off = static_cast<uint32_t>(P.mf_contrib_ptr.size());
      M.contrib_off = static_cast<uint32_t>(P.mf_contrib_idx.size());
      M.cent_off = static_cast<uint32_t>(P.mf_cent.size());
      uint32_t count = 0, n_cent = 0;
      for (uint32_t e = 0; e < task_nent[t]; ++e) {
        const uint32_t ge = task_ent_base[t] + e;
        if (mc_ptr[ge + 1] == mc_ptr[ge]) continue;
        P.mf_cent.push_back(static_cast<uint16_t>(e));
        P.mf_contrib_ptr.push_back(count);
        P.mf_contrib_idx.insert(P.mf_contrib_idx.end(), mc.begin() + mc_ptr[ge], mc.begin() + mc_ptr[ge + 1]);
        count += mc_ptr[ge + 1] - mc_ptr[ge];
        ++n_cent;
      }
      P.mf_contrib_ptr.push_back(count);
      M.n_cent = n_cent;
      M.n_contrib_idx = count;
      while (P.mf_contrib_idx.size() % 4) P.mf_contrib_idx.push_back(0);
    }
    P.mf = ok;
    for (int k = 0; k < 16; ++k) {
      P.mf_tab.push_back(0);
      P.mf_ext.push_back(0);
      P.mf_contrib_ptr.push_back(0);
      P.mf_contrib_idx.push_back(0);
      P.mf_cent.push_back(0);
      P.mf_anc.push_back(0);
      P.mf_lvl_ptr.push_back(0);
      P.mf_fronts.push_back(LdltFront{});
    }
    if (std::getenv("SLPX_LDLT_VERBOSE")) {
      size_t tab = 0, arena = 0;
      for (auto& M : P.mf_tasks) {
        tab = std::max<size_t>(tab, M.n_tab);
        arena = std::max<size_t>(arena, M.arena);
      }
      {
        std::map<std::pair<int, int>, int> hist;
        std::map<int, int> rows;
        for (size_t i = 0; i + 16 < P.mf_fronts.size(); ++i) {
          ++hist[{P.mf_fronts[i].w, P.mf_fronts[i].nch}];
          ++rows[P.mf_fronts[i].nr - P.mf_fronts[i].w - 1];
        }
        std::fprintf(stderr, "ldlt fronts (w, children values per entry): count —");
        for (auto& [k, v] : hist) std::fprintf(stderr, " (%d,%d):%d", k.first, k.second, v);
        std::fprintf(stderr, "\nldlt fronts rows below the pivots: count —");
        for (auto& [k, v] : rows) std::fprintf(stderr, " %d:%d", k, v);
        std::fprintf(stderr, "\n");
      }
      for (int r = 0; r < P.n_rounds && ok; ++r) {
        uint32_t most = 0, over = 0, lv = 0, deepest = 0, most_passes = 0, least_passes = 1u << 30;
        for (uint32_t ti = P.round_ptr[r]; ti < P.round_ptr[r + 1]; ++ti) {
          const LdltTask& T = P.tasks[ti];
          deepest = std::max(deepest, T.n_lvl);
          uint32_t passes = 0;
          for (uint32_t l = 0; l < T.n_lvl; ++l) passes += (P.mf_lvl_ptr[T.lvl_off + l + 1] - P.mf_lvl_ptr[T.lvl_off + l] + 15u) / 16u;
          most_passes = std::max(most_passes, passes);
          least_passes = std::min(least_passes, passes);
          for (uint32_t l = 0; l < T.n_lvl; ++l) {
            const uint32_t nf = P.mf_lvl_ptr[T.lvl_off + l + 1] - P.mf_lvl_ptr[T.lvl_off + l];
            most = std::max(most, nf);
            over += nf > 16;
            ++lv;
          }
        }
        std::fprintf(stderr, "ldlt fronts round %d: deepest task %u levels, most fronts in a level %u, levels with more than 16 fronts %u of %u; passes of 16 waves per task %u..%u\n",
                     r, deepest, most, over, lv, least_passes, most_passes);
      }
      if (const char* lv = std::getenv("SLPX_LDLT_VERBOSE"); lv != nullptr && std::atoi(lv) >= 2) {
        // per task: round, levels, and per level "fronts:widest w" — where the critical path's imbalance sits
        for (size_t ti = 0; ti < P.tasks.size() && ok; ++ti) {
          const LdltTask& T = P.tasks[ti];
          const LdltMfTask& M = P.mf_tasks[ti];
          std::fprintf(stderr, "ldlt task %zu round %u cols %u ents %u anc %u levels %u:", ti, T.round, T.n_col, T.n_ent, M.n_anc, T.n_lvl);
          for (uint32_t l = 0; l < T.n_lvl; ++l) {
            const uint32_t b = P.mf_lvl_ptr[T.lvl_off + l], e = P.mf_lvl_ptr[T.lvl_off + l + 1];
            uint32_t wmax = 0, nchmax = 0, rmax = 0;
            for (uint32_t q = b; q < e; ++q) {
              const LdltFront& F = P.mf_fronts[M.front_off + q];
              wmax = std::max<uint32_t>(wmax, F.w);
              nchmax = std::max<uint32_t>(nchmax, F.nch);
              rmax = std::max<uint32_t>(rmax, F.nr - F.w - 1u);
            }
            std::fprintf(stderr, " %u:w%u,c%u,r%u", e - b, wmax, nchmax, rmax);
          }
          std::fprintf(stderr, "\n");
          if (std::atoi(lv) >= 3 && T.round >= 1)
            for (uint32_t l = 0; l < T.n_lvl; ++l)
              for (uint32_t q = P.mf_lvl_ptr[T.lvl_off + l]; q < P.mf_lvl_ptr[T.lvl_off + l + 1]; ++q) {
                const LdltFront& F = P.mf_fronts[M.front_off + q];
                std::fprintf(stderr, "  level %u front w%u r%u nch%u original columns:", l, F.w, F.nr - F.w - 1u, F.nch);
                for (uint32_t c = 0; c < F.w; ++c) std::fprintf(stderr, " %d", P.perm[P.col_perm[T.col_off + F.col0 + c]]);
                std::fprintf(stderr, "\n");
              }
        }
      }
      std::fprintf(stderr, "ldlt multifrontal plan: %s, %zu fronts, widest table %zu bytes, largest arena %zu doubles, most children per entry %u, update slots %u (pair plan: %u), fronts on the matrix cores %u\n",
                   ok ? "built" : "NOT built", P.mf_fronts.size() - 16, 2 * tab, arena, P.mf_max_nch, P.mf_n_contrib, P.n_contrib, P.mf_n_mfma);
    }
  } catch (const MfRefused& refused) {
    // (a consistency check of the fronts tripped: the pair-list plan takes over — a slower step, never a wrong one —
    // and says so once, whatever the verbosity: a defect of the symbolic phase must not hide as a slow-down)
    static std::atomic<bool> warned{false};
    if (!warned.exchange(true) || std::getenv("SLPX_LDLT_VERBOSE"))
      std::fprintf(stderr, "slpx: the multifrontal plan of an order-%d system was refused (%s); the pair-list plan is used\n", n, refused.why);
    P.mf = false;
    P.mf_tasks.assign(ntasks, LdltMfTask{});
    P.mf_n_contrib = P.mf_max_nch = P.mf_max_front_rows = P.mf_n_mfma = 0;
    // (the tail padding the staged kernels' 16-byte reads expect, as above)
    P.mf_tab.assign(16, 0);
    P.mf_ext.assign(16, 0);
    P.mf_contrib_ptr.assign(16, 0);
    P.mf_contrib_idx.assign(16, 0);
    P.mf_cent.assign(16, 0);
    P.mf_anc.assign(16, 0);
    P.mf_lvl_ptr.assign(16, 0);
    P.mf_fronts.assign(16, LdltFront{});
  }

  lap("  ldlt: fronts and their tables");
  // structural zero pivots of the unregularized matrix: a diagonal of the (2,2)
  // block (no lhs source other than the forced 0) that no earlier column updates
  for (int j = 0; j < n; ++j)
    if (!has_diag[P.perm[j]] && !diag_updated_exact[j]) P.structurally_singular_unregularized = true;
  if (ordering_forced) P.structurally_singular_unregularized = true;

  // tail padding: the staged kernels read whole 16-byte groups
  for (int k = 0; k < 16; ++k) {
    P.ent_src.push_back(-1);
    P.ent_flags.push_back(0);
    P.ent_col.push_back(0);
    P.ent_out.push_back(0);
    P.pairs.push_back({});
    P.ent_pair_ptr.push_back(0);
    P.ent_contrib_ptr.push_back(0);
    P.lvl_ptr.push_back(0);
    P.col_lvl_ptr.push_back(0);
    P.col_perm.push_back(0);
    P.fwd_ptr.push_back(0);
    P.fwd_contrib_ptr.push_back(0);
    P.bwd_ptr.push_back(0);
    P.fwd_items.push_back({});
    P.bwd_items.push_back({});
    P.sn_desc.push_back({});
    P.col_sn.push_back(0);
    P.sn_lvl_ptr.push_back(0);
  }
  if (std::getenv("SLPX_LDLT_VERBOSE")) {
    // per round: tasks, levels, supernodes (w >= 2) per level, rows per level
    for (int r = 0; r < P.n_rounds; ++r) {
      std::vector<int> hist(9, 0);
      uint32_t max_sn = 0, max_rows = 0, lvls = 0, ntask = P.round_ptr[r + 1] - P.round_ptr[r];
      double sum_sn = 0, sum_ent = 0;
      for (uint32_t ti = P.round_ptr[r]; ti < P.round_ptr[r + 1]; ++ti) {
        const LdltTask& T = P.tasks[ti];
        for (uint32_t l = 0; l < T.n_lvl; ++l) {
          const uint32_t ns = P.sn_lvl_ptr[T.lvl_off + l + 1] - P.sn_lvl_ptr[T.lvl_off + l];
          uint32_t ni = 0;
          for (uint32_t q = P.sn_lvl_ptr[T.lvl_off + l]; q < P.sn_lvl_ptr[T.lvl_off + l + 1]; ++q) ni += P.sn_desc[T.sn_off + q].nr;
          const uint32_t ne = P.lvl_ptr[T.lvl_off + l + 1] - P.lvl_ptr[T.lvl_off + l];
          max_sn = std::max(max_sn, ns);
          max_rows = std::max(max_rows, ni);
          sum_sn += ns;
          sum_ent += ne;
          ++lvls;
          ++hist[std::min<uint32_t>(8, (ns + 3) / 4)];
        }
      }
      {
        // update blocks for the parents: how many entries per task, how long their pair lists,
        // and how many update slots an entry of this round's tasks takes from its children
        double ext = 0, ext_pairs = 0, own_pairs = 0, own_ent = 0, takes = 0, takers = 0;
        uint32_t ext_max = 0, take_max = 0;
        for (uint32_t ti = P.round_ptr[r]; ti < P.round_ptr[r + 1]; ++ti) {
          const LdltTask& T = P.tasks[ti];
          ext += T.n_ext;
          ext_max = std::max(ext_max, T.n_ext);
          const uint32_t* pp = P.ent_pair_ptr.data() + T.pair_ptr_off;
          ext_pairs += pp[T.n_ent + T.n_ext] - pp[T.n_ent];
          own_pairs += pp[T.n_ent];
          own_ent += T.n_ent;
          const uint32_t* cp = P.ent_contrib_ptr.data() + T.contrib_ptr_off;
          for (uint32_t i = 0; i < T.n_ent; ++i)
            if (cp[i + 1] > cp[i]) {
              takes += cp[i + 1] - cp[i];
              takers += 1;
              take_max = std::max(take_max, cp[i + 1] - cp[i]);
            }
        }
        std::fprintf(stderr,
                     "ldlt round %d: update entries/task avg %.0f max %u, pairs per update entry %.1f, pairs per own entry "
                     "%.1f; entries taking update slots/task %.0f, slots per such entry avg %.1f max %u\n",
                     r, ext / ntask, ext_max, ext > 0 ? ext_pairs / ext : 0.0, own_pairs / std::max(1.0, own_ent),
                     takers / ntask, takers > 0 ? takes / takers : 0.0, take_max);
      }
      std::fprintf(stderr, "ldlt round %d: %u tasks, %.1f levels/task, chains(w>=2)/level avg %.1f max %u, chain rows/level max %u, entries/level avg %.0f; chains/level hist (0,1-4,5-8,...,>=29):",
                   r, ntask, double(lvls) / ntask, sum_sn / lvls, max_sn, max_rows, sum_ent / lvls);
      for (int h : hist) std::fprintf(stderr, " %d", h);
      std::fprintf(stderr, "\n");
    }
  }
  for (int r = 0; r < P.n_rounds; ++r) {
    uint32_t deepest = 0;
    for (uint32_t ti = P.round_ptr[r]; ti < P.round_ptr[r + 1]; ++ti) deepest = std::max(deepest, P.tasks[ti].n_lvl);
    P.critical_levels += static_cast<int>(deepest);
  }
  if (std::getenv("SLPX_LDLT_VERBOSE")) std::fprintf(stderr, "ldlt LDS: factor %u bytes, solve %u bytes\n", P.factor_lds_bytes, P.solve_lds_bytes);
  if (P.factor_lds_bytes > 160u * 1024u || P.solve_lds_bytes > 160u * 1024u)
    throw std::runtime_error("ldlt: a task's working set exceeds the 160 KB LDS of a CU; lower "
                             "LdltOptions::task_entries");
  P.flops = 2 * static_cast<int64_t>(n_real_pairs);
  P.factor_bytes = 12LL * lower.nnz() + 16LL * (P.nnzL + n);
  P.solve_bytes = 32LL * P.nnzL + 16LL * n;
  return P;
}

bool ldlt_plan_error_is_too_big(const std::exception& e) {
  const std::string what = e.what();
  return what.find("exceeds the LDS task budget") != std::string::npos || what.find("exceeds the 160 KB LDS") != std::string::npos ||
         what.find("exceeds 16-bit local indexing") != std::string::npos;
}

LdltPlan build_dense_ldlt_plan(const CscPattern& lower, int n_dec) {
  LdltPlan P;
  const int n = lower.cols;
  P.n = n;
  P.n_dec = n_dec;
  P.dense = true;
  P.perm.resize(n);
  std::iota(P.perm.begin(), P.perm.end(), 0);
  P.iperm = P.perm;
  P.parent.assign(n, -1);
  for (int j = 0; j + 1 < n; ++j) P.parent[j] = j + 1;
  P.nnzL = static_cast<int64_t>(n) * (n - 1) / 2;
  P.Lp.assign(n + 1, 0);
  for (int j = 0; j < n; ++j) P.Lp[j + 1] = P.Lp[j] + (n - 1 - j);
  P.Li.reserve(static_cast<size_t>(P.nnzL));
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) P.Li.push_back(i);
  P.etree_height = n;
  P.n_rounds = 0;
  P.round_ptr.assign(1, 0);
  P.n_supernodes = n;
  P.critical_levels = n;
  // (the padding the uploads and the staged kernels' 16-byte reads expect of every plan)
  for (int k = 0; k < 16; ++k) {
    P.ent_src.push_back(-1);
    P.ent_flags.push_back(0);
    P.ent_col.push_back(0);
    P.ent_out.push_back(0);
    P.pairs.push_back({});
    P.ent_pair_ptr.push_back(0);
    P.ent_contrib_ptr.push_back(0);
    P.contrib_idx.push_back(0);
    P.ext_dst.push_back(0);
    P.lvl_ptr.push_back(0);
    P.col_lvl_ptr.push_back(0);
    P.col_perm.push_back(0);
    P.fwd_ptr.push_back(0);
    P.fwd_contrib_ptr.push_back(0);
    P.scontrib_idx.push_back(0);
    P.bwd_ptr.push_back(0);
    P.fwd_items.push_back({});
    P.bwd_items.push_back({});
    P.sext_ptr.push_back(0);
    P.sext_dst.push_back(0);
    P.sext_items.push_back({});
    P.sn_desc.push_back({});
    P.col_sn.push_back(0);
    P.sn_lvl_ptr.push_back(0);
    P.mf_tab.push_back(0);
    P.mf_ext.push_back(0);
    P.mf_contrib_ptr.push_back(0);
    P.mf_contrib_idx.push_back(0);
    P.mf_cent.push_back(0);
    P.mf_anc.push_back(0);
    P.mf_lvl_ptr.push_back(0);
    P.mf_fronts.push_back(LdltFront{});
  }
  // the traffic of a dense factorization and of its two triangular solves
  P.factor_bytes = 12LL * lower.nnz() + 16LL * (P.nnzL + n);
  P.solve_bytes = 32LL * P.nnzL + 16LL * n;
  P.flops = static_cast<int64_t>(n) * n * n / 3;
  return P;
}

LdltPlan build_ldlt_plan(const CscPattern& lower, int n_dec, const LdltOptions& opt,
                         const std::vector<int32_t>* user_perm,
                         const std::vector<uint8_t>* diag_has_source) {
  auto too_big = [](const std::runtime_error& e) { return ldlt_plan_error_is_too_big(e); };
  try {
    return build_ldlt_plan_once(lower, n_dec, opt, user_perm, diag_has_source);
  } catch (const std::runtime_error& e) {
    if (!too_big(e)) throw;
  }
  // A column that does not fit a task is usually the work of a few well-connected nodes the
  // default rule did not take for hubs: once more with a sharper rule; then with bigger tasks
  // (a column of c entries needs about c^2 / 8 entry-equivalents: 2048 hold c = 124, 5120 c = 198;
  // beyond that the factor is dense enough for the reference's dense branch).
  LdltOptions sharper = opt;
  if (user_perm == nullptr || user_perm->empty()) {
    sharper.hub_factor = 2.0;
    sharper.hub_floor = 12;
  }
  for (uint32_t entries : {std::max<uint32_t>(opt.task_entries, LdltOptions{}.task_entries), 4096u, 5120u}) {
    if (entries < opt.task_entries) continue;
    sharper.task_entries = entries;
    try {
      return build_ldlt_plan_once(lower, n_dec, sharper, user_perm, diag_has_source);
    } catch (const std::runtime_error& e) {
      if (!too_big(e) || entries == 5120u) throw;
    }
  }
  throw std::runtime_error("ldlt: no plan");  // not reached
}

}  // namespace slpx
