// NLP structure: everything Problem::solve sets up once before the interior-point
// loop in the reference (include/sleipnir/optimization/problem.hpp:517-560):
//   Gradient g{f, x}; Hessian<Lower> H_f{f, x}; Hessian<Lower> H_c{-yᵀc_e - zᵀc_i, x};
//   Jacobian A_e{c_e, x}; Jacobian A_i{c_i, x}
// re-expressed for the device: static CSC patterns, one value vector V with a
// fixed slot per structural nonzero, cached values for LINEAR rows
// (jacobian.hpp:89-94, hessian.hpp:84-89) and two tape programs — `full`
// (values + every nonlinear derivative row) and `values` (f, c_e, c_i only, for
// line-search trial points, interior_point.hpp:513-527).
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

#include "graph.hpp"
#include "tape_compiler.hpp"

namespace slpx {

struct CscPattern {
  int rows = 0, cols = 0;
  std::vector<int32_t> colptr, rowidx;
  int nnz() const { return static_cast<int>(rowidx.size()); }
};

struct NlpStructure {
  int n = 0, m_e = 0, m_i = 0;
  // V layout: [f | c_e | c_i | g | A_e | A_i | H_f | H_c]
  int off_f = 0, off_ce = 0, off_ci = 0, off_g = 0, off_Ae = 0, off_Ai = 0, off_Hf = 0, off_Hc = 0;
  int nV = 0;
  CscPattern g_pat;  // 1 x n (colptr over n columns)
  CscPattern Ae, Ai;
  CscPattern Hf, Hc;  // lower triangles, n x n
  std::vector<double> V_static_raw;  // unscaled cached values (linear rows), 0 elsewhere
  std::vector<int32_t> V_scale_idx;  // scale index of every V entry (-1 = none)
  std::vector<uint8_t> V_is_static;  // 1 = never written by the tape
  TapeProgram full, values;
  // Separable sums cut into partial sums (nlp.cpp): V[dst] = scale[scale_idx] * sum of the
  // `count` partials stored in the hidden tail of V starting at src_off; executed by
  // tape_reduce_kernel after every sweep.
  struct SumReduce {
    int32_t dst, scale_idx, src_off, count;
  };
  std::vector<SumReduce> reduces;
  // expression types (problem.hpp:236-263)
  uint8_t f_type = T_NONE, ce_type = T_NONE, ci_type = T_NONE;
  // leaf nodes created for the duals (problem.hpp:519-520)
  std::vector<NodeId> y_nodes, z_nodes;
  // bookkeeping for reports
  size_t graph_nodes_before = 0, graph_nodes_after = 0;
  int nonlinear_rows = 0, linear_rows = 0;

  int n_inputs() const { return n + m_e + m_i; }
  int n_scales() const { return 1 + m_e + m_i; }
};

// Builds the structure.  `x` are the decision-variable leaf nodes in problem
// order (problem.hpp:96-100), `f` the cost root (kNull = no cost), `c_e`/`c_i`
// the constraint roots already in `lhs - rhs` form (variable.hpp:716-778).
NlpStructure build_nlp_structure(Graph& g, const std::vector<NodeId>& x, NodeId f,
                                 const std::vector<NodeId>& c_e, const std::vector<NodeId>& c_i,
                                 const TapeCompileOptions& opt = {},
                                 const std::function<void(const NlpStructure&)>& on_patterns = {});
// (`on_patterns` runs on a thread of its own as soon as everything but the two tape programs is
// in place — sizes, V layout, sparsity patterns — and is joined before the function returns: the
// KKT plan and the symbolic factorization only need those, and the tape compiler is the long pole)

}  // namespace slpx
