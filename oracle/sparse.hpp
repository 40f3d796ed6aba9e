// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// Minimal CSC sparse algebra standing in for the Eigen routines the reference
// calls on this path.  Eigen is a FetchContent dependency of the reference
// (CMakeLists.txt:77-84, GIT_TAG c92d9c379dd034d3e7ddb7cdd7ba0add3bf1c747) and is
// NOT present under /root/reference, so its published behaviour is restated:
//   * setFromTriplets: column-major compressed, inner indices sorted, duplicate
//     (row,col) summed, explicit zeros kept (call sites jacobian.hpp:103,153,
//     hessian.hpp:98,151; interior_point.hpp:440 setFromSortedTriplets)
//   * sparse + sparse = union pattern; sparse products are structural (no pruning)
//   * triangularView<Lower> keeps row >= col
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <vector>

#include "ad.hpp"

namespace orc {

struct CSC {
  int rows = 0, cols = 0;
  std::vector<int> colptr;  // cols + 1
  std::vector<int> rowidx;
  std::vector<double> val;

  CSC() : colptr(1, 0) {}
  CSC(int r, int c) : rows(r), cols(c), colptr(c + 1, 0) {}
  int nnz() const { return static_cast<int>(rowidx.size()); }
};

using Vec = std::vector<double>;

inline CSC from_triplets(int rows, int cols, const std::vector<Triplet>& t) {
  CSC m(rows, cols);
  std::vector<int> count(cols + 1, 0);
  for (const auto& e : t) ++count[e.col + 1];
  for (int c = 0; c < cols; ++c) count[c + 1] += count[c];
  std::vector<int> ri(t.size());
  std::vector<double> va(t.size());
  {
    std::vector<int> next(count.begin(), count.end() - 1);
    for (const auto& e : t) {
      int p = next[e.col]++;
      ri[p] = e.row;
      va[p] = e.value;
    }
  }
  for (int c = 0; c < cols; ++c) {
    int b = count[c], e = count[c + 1];
    std::vector<std::pair<int, double>> col;
    col.reserve(e - b);
    for (int p = b; p < e; ++p) col.emplace_back(ri[p], va[p]);
    std::stable_sort(col.begin(), col.end(),
                     [](const auto& a, const auto& b2) { return a.first < b2.first; });
    for (size_t k = 0; k < col.size(); ++k) {
      if (!m.rowidx.empty() && static_cast<int>(m.rowidx.size()) > m.colptr[c] &&
          m.rowidx.back() == col[k].first) {
        m.val.back() += col[k].second;
      } else {
        m.rowidx.push_back(col[k].first);
        m.val.push_back(col[k].second);
      }
    }
    m.colptr[c + 1] = static_cast<int>(m.rowidx.size());
  }
  return m;
}

inline CSC transpose(const CSC& a) {
  std::vector<Triplet> t;
  t.reserve(a.nnz());
  for (int c = 0; c < a.cols; ++c)
    for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) t.push_back({c, a.rowidx[p], a.val[p]});
  return from_triplets(a.cols, a.rows, t);
}

// alpha*A + beta*B with union pattern
inline CSC add(const CSC& a, const CSC& b, double alpha = 1.0, double beta = 1.0) {
  assert(a.rows == b.rows && a.cols == b.cols);
  CSC m(a.rows, a.cols);
  for (int c = 0; c < a.cols; ++c) {
    int pa = a.colptr[c], ea = a.colptr[c + 1];
    int pb = b.colptr[c], eb = b.colptr[c + 1];
    while (pa < ea || pb < eb) {
      if (pb >= eb || (pa < ea && a.rowidx[pa] < b.rowidx[pb])) {
        m.rowidx.push_back(a.rowidx[pa]);
        m.val.push_back(alpha * a.val[pa]);
        ++pa;
      } else if (pa >= ea || b.rowidx[pb] < a.rowidx[pa]) {
        m.rowidx.push_back(b.rowidx[pb]);
        m.val.push_back(beta * b.val[pb]);
        ++pb;
      } else {
        m.rowidx.push_back(a.rowidx[pa]);
        m.val.push_back(alpha * a.val[pa] + beta * b.val[pb]);
        ++pa;
        ++pb;
      }
    }
    m.colptr[c + 1] = static_cast<int>(m.rowidx.size());
  }
  return m;
}

inline CSC scaled(const CSC& a, double alpha) {
  CSC m = a;
  for (auto& v : m.val) v *= alpha;
  return m;
}

// diag(d) * A  (row scaling)
inline CSC row_scaled(const Vec& d, const CSC& a) {
  CSC m = a;
  for (int p = 0; p < m.nnz(); ++p) m.val[p] *= d[m.rowidx[p]];
  return m;
}

inline CSC lower_triangle(const CSC& a) {
  CSC m(a.rows, a.cols);
  for (int c = 0; c < a.cols; ++c) {
    for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
      if (a.rowidx[p] >= c) {
        m.rowidx.push_back(a.rowidx[p]);
        m.val.push_back(a.val[p]);
      }
    }
    m.colptr[c + 1] = static_cast<int>(m.rowidx.size());
  }
  return m;
}

// Structural sparse product A * B (column-by-column accumulation, sorted output)
inline CSC multiply(const CSC& a, const CSC& b) {
  assert(a.cols == b.rows);
  CSC m(a.rows, b.cols);
  std::vector<double> acc(a.rows, 0.0);
  std::vector<int> mark(a.rows, -1);
  std::vector<int> idx;
  for (int c = 0; c < b.cols; ++c) {
    idx.clear();
    for (int pb = b.colptr[c]; pb < b.colptr[c + 1]; ++pb) {
      int k = b.rowidx[pb];
      double bv = b.val[pb];
      for (int pa = a.colptr[k]; pa < a.colptr[k + 1]; ++pa) {
        int r = a.rowidx[pa];
        if (mark[r] != c) {
          mark[r] = c;
          acc[r] = 0.0;
          idx.push_back(r);
        }
        acc[r] += a.val[pa] * bv;
      }
    }
    std::sort(idx.begin(), idx.end());
    for (int r : idx) {
      m.rowidx.push_back(r);
      m.val.push_back(acc[r]);
    }
    m.colptr[c + 1] = static_cast<int>(m.rowidx.size());
  }
  return m;
}

inline CSC diag_matrix(const Vec& d) {
  int n = static_cast<int>(d.size());
  CSC m(n, n);
  for (int c = 0; c < n; ++c) {
    m.rowidx.push_back(c);
    m.val.push_back(d[c]);
    m.colptr[c + 1] = c + 1;
  }
  return m;
}

// y = A x
inline Vec spmv(const CSC& a, const Vec& x) {
  Vec y(a.rows, 0.0);
  for (int c = 0; c < a.cols; ++c)
    for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) y[a.rowidx[p]] += a.val[p] * x[c];
  return y;
}

// y = Aᵀ x
inline Vec spmv_t(const CSC& a, const Vec& x) {
  Vec y(a.cols, 0.0);
  for (int c = 0; c < a.cols; ++c) {
    double s = 0.0;
    for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) s += a.val[p] * x[a.rowidx[p]];
    y[c] = s;
  }
  return y;
}

inline bool all_finite(const CSC& a) {
  for (double v : a.val)
    if (!std::isfinite(v)) return false;
  return true;
}
inline bool all_finite(const Vec& a) {
  for (double v : a)
    if (!std::isfinite(v)) return false;
  return true;
}

inline double norm_inf(const Vec& v) {
  double m = 0.0;
  for (double x : v) m = std::max(m, std::abs(x));
  return m;
}
inline double norm_1(const Vec& v) {
  double s = 0.0;
  for (double x : v) s += std::abs(x);
  return s;
}
inline double norm_2(const Vec& v) {
  double s = 0.0;
  for (double x : v) s += x * x;
  return std::sqrt(s);
}
inline double dot(const Vec& a, const Vec& b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

// util/sparse_inf_norms.hpp:17-31
inline Vec sparse_inf_norms(const CSC& mat) {
  Vec norms(mat.rows, 0.0);
  for (int p = 0; p < mat.nnz(); ++p)
    norms[mat.rowidx[p]] = std::max(norms[mat.rowidx[p]], std::abs(mat.val[p]));
  return norms;
}

// ---------------------------------------------------------------------------
// jacobian.hpp / hessian.hpp / gradient.hpp
// ---------------------------------------------------------------------------

class Jacobian {
 public:
  Jacobian() = default;
  // jacobian.hpp:54-105
  Jacobian(std::vector<Expr*> variables, std::vector<Expr*> wrt)
      : m_variables(std::move(variables)), m_wrt(std::move(wrt)) {
    init();
  }

  // jacobian.hpp:134-156
  const CSC& value() {
    if (m_nonlinear_rows.empty()) return m_J;
    for (auto& top_list : m_top_lists) update_values(top_list);
    std::vector<Triplet> triplets = m_cached_triplets;
    for (int row : m_nonlinear_rows)
      append_triplets(m_top_lists[row], m_output_lists[row], triplets, row);
    m_J = from_triplets(static_cast<int>(m_variables.size()), static_cast<int>(m_wrt.size()),
                        triplets);
    finish(m_J);
    return m_J;
  }

  int num_nonlinear_rows() const { return static_cast<int>(m_nonlinear_rows.size()); }
  size_t total_list_nodes() const {
    size_t s = 0;
    for (auto& l : m_top_lists) s += l.size();
    return s;
  }

 protected:
  virtual void finish(CSC&) {}

  void init() {
    for (Expr* v : m_variables) m_top_lists.push_back(topological_sort(v));
    for (size_t col = 0; col < m_wrt.size(); ++col) m_wrt[col]->scratch = static_cast<int>(col);
    for (auto& top_list : m_top_lists) {
      m_output_lists.emplace_back();
      for (Expr* node : top_list)
        if (node->scratch != -1) m_output_lists.back().emplace_back(node->scratch, node);
    }
    for (Expr* w : m_wrt) w->scratch = -1;
    for (int row = 0; row < static_cast<int>(m_variables.size()); ++row) {
      if (m_variables[row] == nullptr) continue;
      if (m_variables[row]->type == LINEAR) {
        append_triplets(m_top_lists[row], m_output_lists[row], m_cached_triplets, row);
      } else if (m_variables[row]->type > LINEAR) {
        m_nonlinear_rows.push_back(row);
      }
    }
    if (m_nonlinear_rows.empty()) {
      m_J = from_triplets(static_cast<int>(m_variables.size()), static_cast<int>(m_wrt.size()),
                          m_cached_triplets);
    } else {
      m_J = CSC(static_cast<int>(m_variables.size()), static_cast<int>(m_wrt.size()));
    }
  }

  std::vector<Expr*> m_variables;
  std::vector<Expr*> m_wrt;
  std::vector<Graph> m_top_lists;
  std::vector<OutputList> m_output_lists;
  CSC m_J;
  std::vector<Triplet> m_cached_triplets;
  std::vector<int> m_nonlinear_rows;
};

// hessian.hpp:35-184.  lower == true is Hessian<Scalar, Eigen::Lower>.
class Hessian : public Jacobian {
 public:
  Hessian() = default;
  Hessian(Expr* variable, std::vector<Expr*> wrt, bool lower) : m_lower(lower) {
    m_wrt = std::move(wrt);
    m_variables = gradient_tree(topological_sort(variable), m_wrt);  // hessian.hpp:50-51
    init();
    if (m_nonlinear_rows.empty()) finish(m_J);  // hessian.hpp:97-102
  }

 protected:
  void finish(CSC& m) override {
    if (m_lower) m = lower_triangle(m);  // hessian.hpp:152-154
  }
  bool m_lower = false;
};

// gradient.hpp:25-66: 1-row Jacobian; value() as a dense vector here
class Gradient {
 public:
  Gradient() = default;
  Gradient(Expr* variable, std::vector<Expr*> wrt) : m_jac({variable}, std::move(wrt)) {}
  // Sparse row as a 1 x n CSC
  const CSC& value() { return m_jac.value(); }
  Jacobian& jacobian() { return m_jac; }

 private:
  Jacobian m_jac;
};

inline Vec sparse_row_to_dense(const CSC& row) {
  Vec g(row.cols, 0.0);
  for (int c = 0; c < row.cols; ++c)
    for (int p = row.colptr[c]; p < row.colptr[c + 1]; ++p) g[c] += row.val[p];
  return g;
}

}  // namespace orc
