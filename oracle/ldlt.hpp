// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// Restates
//   include/sleipnir/optimization/solver/util/inertia.hpp:40-50
//   include/sleipnir/optimization/solver/util/sparse_regularized_ldlt.hpp:64-224
//   include/sleipnir/optimization/solver/util/dense_regularized_ldlt.hpp:59-136
//   include/sleipnir/optimization/solver/util/regularized_ldlt.hpp:17-134
// plus stand-ins for the two Eigen solvers those files instantiate
// (Eigen::SimplicialLDLT<SparseMatrix<double>> — Lower, AMDOrdering — and
// Eigen::LDLT<MatrixXd>).  Eigen @ c92d9c37 is not on disk, so
// these are textbook restatements of its published algorithms:
//   * sparse: fill-reducing symmetric permutation (exact minimum degree here, in
//     place of Eigen's approximate minimum degree), elimination tree, up-looking
//     LDLᵀ without pivoting (T. Davis, "Algorithm 849: LDL"), failure
//     (NumericalIssue) on an exactly-zero pivot.
//   * dense: LDLᵀ with symmetric diagonal pivoting on the largest |diagonal|.
// PARITY UNPINNED at this seam: the reference holds no test of RegularizedLDLT
// (SURVEY.md §8c); the seam is pinned only transitively by whole-solve tests.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <set>
#include <vector>

#include "sparse.hpp"

namespace orc {

// inertia.hpp:14-50
struct Inertia {
  int positive = 0, negative = 0, zero = 0;
  Inertia() = default;
  Inertia(int p, int n, int z) : positive(p), negative(n), zero(z) {}
  explicit Inertia(const Vec& D) {
    const double eps = std::numeric_limits<double>::epsilon();
    for (double e : D) {
      if (e > eps) ++positive;
      else if (e < -eps) ++negative;
      else ++zero;
    }
  }
  bool operator==(const Inertia& o) const {
    return positive == o.positive && negative == o.negative && zero == o.zero;
  }
};

enum Info { Success = 0, NumericalIssue = 1 };

// Exact minimum (external) degree ordering on the graph of A + Aᵀ; ties broken
// by lowest index.  Returns perm with perm[new] = old.
inline std::vector<int> minimum_degree_ordering(const CSC& lower) {
  int n = lower.cols;
  std::vector<std::vector<int>> adj(n);
  for (int c = 0; c < n; ++c)
    for (int p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
      int r = lower.rowidx[p];
      if (r != c) {
        adj[r].push_back(c);
        adj[c].push_back(r);
      }
    }
  for (auto& a : adj) {
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
  }
  std::set<std::pair<int, int>> pq;
  for (int i = 0; i < n; ++i) pq.insert({static_cast<int>(adj[i].size()), i});
  std::vector<char> gone(n, 0);
  std::vector<int> perm;
  perm.reserve(n);
  std::vector<int> merged;
  while (!pq.empty()) {
    auto [deg, v] = *pq.begin();
    pq.erase(pq.begin());
    gone[v] = 1;
    perm.push_back(v);
    const auto& nv = adj[v];
    for (int u : nv) {
      pq.erase({static_cast<int>(adj[u].size()), u});
      merged.clear();
      std::set_union(adj[u].begin(), adj[u].end(), nv.begin(), nv.end(),
                     std::back_inserter(merged));
      merged.erase(std::remove_if(merged.begin(), merged.end(),
                                  [&](int w) { return w == u || w == v; }),
                   merged.end());
      adj[u] = merged;
      pq.insert({static_cast<int>(adj[u].size()), u});
    }
    adj[v].clear();
    adj[v].shrink_to_fit();
  }
  return perm;
}

// Stand-in for Eigen::SimplicialLDLT<SparseMatrix<double>, Lower, AMDOrdering>.
class SimplicialLDLT {
 public:
  // If `perm` is non-empty it is used instead of the built-in ordering (the
  // parity tests feed the product's permutation through here).
  void set_permutation(std::vector<int> perm) { m_user_perm = std::move(perm); }

  // analyzePattern: ordering + etree + column counts of the permuted matrix
  void analyze_pattern(const CSC& lower) {
    n = lower.cols;
    perm = m_user_perm.empty() ? minimum_degree_ordering(lower) : m_user_perm;
    pinv.assign(n, 0);
    for (int k = 0; k < n; ++k) pinv[perm[k]] = k;
    build_upper(lower);
    parent.assign(n, -1);
    std::vector<int> flag(n), lnz(n, 0);
    for (int k = 0; k < n; ++k) {
      flag[k] = k;
      for (int p = Up[k]; p < Up[k + 1]; ++p) {
        int i = Ui[p];
        if (i < k) {
          for (; flag[i] != k; i = parent[i]) {
            if (parent[i] == -1) parent[i] = k;
            ++lnz[i];
            flag[i] = k;
          }
        }
      }
    }
    Lp.assign(n + 1, 0);
    for (int k = 0; k < n; ++k) Lp[k + 1] = Lp[k] + lnz[k];
    Li.assign(Lp[n], 0);
    Lx.assign(Lp[n], 0.0);
    D.assign(n, 0.0);
    analyzed = true;
  }

  // factorize: numeric up-looking LDLᵀ on the analysed pattern
  Info factorize(const CSC& lower) {
    build_upper(lower);  // values (pattern is identical by contract)
    std::vector<double> y(n, 0.0);
    std::vector<int> pattern(n), flag(n), lnz(n, 0);
    info = Success;
    for (int k = 0; k < n; ++k) {
      y[k] = 0.0;
      int top = n;
      flag[k] = k;
      for (int p = Up[k]; p < Up[k + 1]; ++p) {
        int i = Ui[p];
        if (i <= k) {
          y[i] += Ux[p];
          int len = 0;
          for (; flag[i] != k; i = parent[i]) {
            pattern[len++] = i;
            flag[i] = k;
          }
          while (len > 0) pattern[--top] = pattern[--len];
        }
      }
      double d = y[k];
      y[k] = 0.0;
      for (; top < n; ++top) {
        int i = pattern[top];
        double yi = y[i];
        y[i] = 0.0;
        int p2 = Lp[i] + lnz[i];
        for (int p = Lp[i]; p < p2; ++p) y[Li[p]] -= Lx[p] * yi;
        double l_ki = yi / D[i];
        d -= l_ki * yi;
        Li[p2] = k;
        Lx[p2] = l_ki;
        ++lnz[i];
      }
      D[k] = d;
      if (d == 0.0) {
        info = NumericalIssue;  // Eigen: zero pivot stops the factorization
        return info;
      }
    }
    return info;
  }

  Vec solve(const Vec& b) const {
    Vec x(n);
    for (int k = 0; k < n; ++k) x[k] = b[perm[k]];
    for (int j = 0; j < n; ++j)
      for (int p = Lp[j]; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * x[j];
    for (int j = 0; j < n; ++j) x[j] /= D[j];
    for (int j = n - 1; j >= 0; --j)
      for (int p = Lp[j]; p < Lp[j + 1]; ++p) x[j] -= Lx[p] * x[Li[p]];
    Vec out(n);
    for (int k = 0; k < n; ++k) out[perm[k]] = x[k];
    return out;
  }

  const Vec& vectorD() const { return D; }
  int nnzL() const { return Lp.empty() ? 0 : Lp[n]; }

  int n = 0;
  bool analyzed = false;
  Info info = Success;
  std::vector<int> perm, pinv, parent, Lp, Li;
  Vec Lx, D;

 private:
  // Upper triangle (CSC) of P A Pᵀ built from the lower triangle of A.
  void build_upper(const CSC& lower) {
    std::vector<int> count(n + 1, 0);
    for (int c = 0; c < n; ++c)
      for (int p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
        int i = pinv[lower.rowidx[p]], j = pinv[c];
        ++count[std::max(i, j) + 1];
      }
    for (int c = 0; c < n; ++c) count[c + 1] += count[c];
    Up = count;
    Ui.assign(count[n], 0);
    Ux.assign(count[n], 0.0);
    std::vector<int> next(count.begin(), count.end() - 1);
    for (int c = 0; c < n; ++c)
      for (int p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
        int i = pinv[lower.rowidx[p]], j = pinv[c];
        int col = std::max(i, j), row = std::min(i, j);
        int q = next[col]++;
        Ui[q] = row;
        Ux[q] = lower.val[p];
      }
  }
  std::vector<int> Up, Ui;
  Vec Ux;
  std::vector<int> m_user_perm;
};

// Stand-in for Eigen::LDLT<MatrixXd> (diagonal pivoting).  `a` is a dense
// column-major n x n matrix of which only the lower triangle is read.
class DenseLDLT {
 public:
  Info compute(std::vector<double> a, int n_) {
    n = n_;
    m = std::move(a);
    trans.assign(n, 0);
    bool ret = true;
    bool found_zero_pivot = false;
    std::vector<double> temp(n);
    auto at = [&](int r, int c) -> double& { return m[static_cast<size_t>(c) * n + r]; };
    for (int k = 0; k < n; ++k) {
      int big = k;
      double bigv = std::abs(at(k, k));
      for (int j = k + 1; j < n; ++j)
        if (std::abs(at(j, j)) > bigv) {
          bigv = std::abs(at(j, j));
          big = j;
        }
      trans[k] = big;
      if (big != k) {
        int s = n - big - 1;
        for (int c = 0; c < k; ++c) std::swap(at(k, c), at(big, c));
        for (int r = 0; r < s; ++r) std::swap(at(big + 1 + r, k), at(big + 1 + r, big));
        for (int i = k + 1; i < big; ++i) std::swap(at(i, k), at(big, i));
        std::swap(at(k, k), at(big, big));
      }
      int rs = n - k - 1;
      if (k > 0) {
        for (int c = 0; c < k; ++c) temp[c] = at(c, c) * at(k, c);
        double s = 0.0;
        for (int c = 0; c < k; ++c) s += at(k, c) * temp[c];
        at(k, k) -= s;
        for (int r = 0; r < rs; ++r) {
          double s2 = 0.0;
          for (int c = 0; c < k; ++c) s2 += at(k + 1 + r, c) * temp[c];
          at(k + 1 + r, k) -= s2;
        }
      }
      double akk = at(k, k);
      bool pivot_is_valid = std::abs(akk) > 0.0;
      if (k == 0 && !pivot_is_valid) {
        for (int j = 0; j < n; ++j) {
          trans[j] = j;
          for (int r = j + 1; r < n; ++r) ret = ret && (at(r, j) == 0.0);
        }
        info = ret ? Success : NumericalIssue;
        return info;
      }
      if (rs > 0 && pivot_is_valid) {
        for (int r = 0; r < rs; ++r) at(k + 1 + r, k) /= akk;
      } else if (rs > 0) {
        for (int r = 0; r < rs; ++r) ret = ret && (at(k + 1 + r, k) == 0.0);
      }
      if (found_zero_pivot && pivot_is_valid) ret = false;
      else if (!pivot_is_valid) found_zero_pivot = true;
    }
    info = ret ? Success : NumericalIssue;
    return info;
  }

  Vec vectorD() const {
    Vec d(n);
    for (int i = 0; i < n; ++i) d[i] = m[static_cast<size_t>(i) * n + i];
    return d;
  }

  Vec solve(const Vec& b) const {
    Vec x = b;
    auto at = [&](int r, int c) { return m[static_cast<size_t>(c) * n + r]; };
    for (int k = 0; k < n; ++k) std::swap(x[k], x[trans[k]]);
    for (int j = 0; j < n; ++j)
      for (int r = j + 1; r < n; ++r) x[r] -= at(r, j) * x[j];
    const double tol = std::numeric_limits<double>::min();
    for (int i = 0; i < n; ++i) {
      double d = at(i, i);
      if (std::abs(d) > tol) x[i] /= d;
      else x[i] = 0.0;
    }
    for (int j = n - 1; j >= 0; --j)
      for (int r = j + 1; r < n; ++r) x[j] -= at(r, j) * x[r];
    for (int k = n - 1; k >= 0; --k) std::swap(x[k], x[trans[k]]);
    return x;
  }

  int n = 0;
  Info info = Success;

 private:
  std::vector<double> m;
  std::vector<int> trans;
};

// regularized_ldlt.hpp + sparse_regularized_ldlt.hpp + dense_regularized_ldlt.hpp
class RegularizedLDLT {
 public:
  RegularizedLDLT(bool use_sparse, int num_decision_variables, int num_equality_constraints,
                  double gamma_min = 1e-10)
      : m_use_sparse(use_sparse),
        m_n(num_decision_variables),
        m_me(num_equality_constraints),
        m_gamma_min(gamma_min),
        ideal(num_decision_variables, num_equality_constraints, 0) {}

  void set_permutation(std::vector<int> perm) { m_sparse.set_permutation(std::move(perm)); }

  Info info() const { return m_info; }
  double hessian_regularization() const { return m_prev_delta; }
  double constraint_jacobian_regularization() const { return m_prev_gamma; }
  int factorizations() const { return m_factorizations; }
  // Back to the state of a freshly constructed object (:45-51) without losing the symbolic
  // analysis — what a new solve() starts from, minus the one-off analyzePattern.
  void forget_regularization() { m_prev_delta = m_prev_gamma = 0.0; }
  const SimplicialLDLT& sparse_solver() const { return m_sparse; }

  // sparse_regularized_ldlt.hpp:64-152 (dense_regularized_ldlt.hpp:59-136 is the
  // same loop over the dense solver)
  RegularizedLDLT& compute(const CSC& lhs) {
    compute_policy(lhs);
    // ORC_TRACE_FACTORIZATIONS=1: attempts per call and the regularization taken, one line each (how often
    // a step needs a second attempt is what the product's twin attempt is sized on)
    const bool trace = std::getenv("ORC_TRACE_FACTORIZATIONS") != nullptr;  // (read every time: a test turns it on mid-process)
    if (trace) std::fprintf(stderr, "orc attempts %d delta %.3e gamma %.3e\n", m_factorizations, m_prev_delta, m_prev_gamma);
    return *this;
  }
  void compute_policy(const CSC& lhs) {
    m_factorizations = 0;
    // lhs + regularization(0, 0): forces the full diagonal into the pattern (:67)
    CSC unreg = add(lhs, regularization(0.0, 0.0));
    if (m_use_sparse && !m_sparse.analyzed) m_sparse.analyze_pattern(unreg);  // :69-72
    m_info = factor(unreg);
    if (m_info == Success) {
      Vec D = vecD();
      bool far = true;
      for (double d : D) far = far && (std::abs(d) >= 1e-4);
      if (Inertia(D) == ideal && far) {  // :82-87
        m_prev_delta = 0.0;
        m_prev_gamma = 0.0;
        return;
      }
    }
    double delta = m_prev_delta == 0.0
                       ? 1e-4
                       : std::max(m_prev_delta / 2.0, std::numeric_limits<double>::epsilon());
    double gamma = m_gamma_min;
    while (true) {
      m_info = factor(add(lhs, regularization(delta, gamma)));
      if (m_info == Success) {
        Inertia inertia(vecD());
        if (inertia == ideal) {
          m_prev_delta = delta;
          m_prev_gamma = gamma;
          return;
        } else if (inertia.zero > 0) {
          if (gamma == 0.0) {
            gamma = 1e-10;
          } else {
            delta *= 10.0;
            gamma *= 10.0;
          }
        } else if (inertia.negative > ideal.negative) {
          delta *= 10.0;
        } else if (inertia.positive > ideal.positive) {
          gamma = gamma == 0.0 ? 1e-10 : gamma * 10.0;
        }
      } else {
        delta *= 10.0;
        gamma = gamma == 0.0 ? 1e-10 : gamma * 10.0;
      }
      if (delta > 1e20 || gamma > 1e20) {
        m_info = NumericalIssue;
        m_prev_delta = delta;
        m_prev_gamma = gamma;
        return;
      }
    }
  }

  Vec solve(const Vec& rhs) const { return m_use_sparse ? m_sparse.solve(rhs) : m_dense.solve(rhs); }

  Vec vecD() const { return m_use_sparse ? m_sparse.vectorD() : m_dense.vectorD(); }

  // :217-224
  CSC regularization(double delta, double gamma) const {
    Vec v(m_n + m_me);
    for (int i = 0; i < m_n; ++i) v[i] = delta;
    for (int i = 0; i < m_me; ++i) v[m_n + i] = -gamma;
    return diag_matrix(v);
  }

 private:
  Info factor(const CSC& a) {
    ++m_factorizations;
    if (m_use_sparse) return m_sparse.factorize(a);
    int n = a.cols;
    std::vector<double> dense(static_cast<size_t>(n) * n, 0.0);
    for (int c = 0; c < n; ++c)
      for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
        dense[static_cast<size_t>(c) * n + a.rowidx[p]] = a.val[p];
      }
    return m_dense.compute(std::move(dense), n);
  }

  bool m_use_sparse;
  int m_n, m_me;
  double m_gamma_min;
  Inertia ideal;
  SimplicialLDLT m_sparse;
  DenseLDLT m_dense;
  Info m_info = Success;
  double m_prev_delta = 0.0, m_prev_gamma = 0.0;
  int m_factorizations = 0;
};

}  // namespace orc
