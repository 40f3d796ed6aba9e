// KKT plan: the static index maps that turn the reference's per-iteration
// sparse algebra (include/sleipnir/optimization/solver/interior_point.hpp:426-448,
// :470-481 and util/append_as_triplets.hpp:26-48) into three gather kernels.
//
//   lhs = [H + tril(AᵢᵀΣAᵢ)  ·]   lower triangle, CSC, plus the forced full
//         [       Aₑ         0]   diagonal of sparse_regularized_ldlt.hpp:67
//   rhs = −[∇f − Aₑᵀy − Aᵢᵀ(−Σcᵢ + μS⁻¹e + z);  cₑ]
//   pˢ = cᵢ − s + Aᵢpˣ,  pᶻ = μS⁻¹e − z − Σpˢ
//
// The reference rebuilds Σ, a sparse triple product, a triplet list and a
// compressed matrix every iteration; the pattern never changes, so here every
// lhs entry knows once and for all which entries of the value vector V (see
// nlp.hpp) it sums.
#pragma once

#include <cstdint>
#include <vector>

#include "nlp.hpp"

namespace slpx {

struct KktPlan {
  int n = 0, m_e = 0, m_i = 0, dim = 0;
  CscPattern lhs;  // dim x dim, lower triangle, every diagonal entry present

  // lhs entry k = sum_{d in [dptr[k], dptr[k+1])} V[dsrc[d]]
  //             + sum_{p in [pptr[k], pptr[k+1])} V[pa[p]] * (z[pr[p]]/s[pr[p]]) * V[pb[p]]
  std::vector<int32_t> dptr, dsrc;
  std::vector<int32_t> pptr, pa, pb, pr;
  // Fast path of the assembly kernel: fast_src[k] >= 0 when entry k is a plain copy of
  // V[fast_src[k]] (one direct source, no product term: every A_e entry and most of H);
  // -1 = structural zero (forced diagonal of the (2,2) block); -2 = general entry.
  std::vector<int32_t> fast_src;

  // rhs (x part): column gathers over A_e and A_i (CSC), g scattered to dense
  std::vector<int32_t> g_src;  // n entries: V index of ∂f/∂x_j or -1
  // A_e, A_i CSC are the NlpStructure patterns; values live at V[off_Ae + p], V[off_Ai + p]

  // back-substitution: A_i in CSR form (row gathers)
  std::vector<int32_t> ai_rowptr, ai_col, ai_src;  // ai_src = V index

  // A_e in CSR form as well (used by the error norms / infeasibility tests)
  std::vector<int32_t> ae_rowptr, ae_col, ae_src;

  // bytes one assemble / rhs pass must move (SURVEY.md §8d formulas, exact counts)
  int64_t assemble_bytes = 0, rhs_bytes = 0;
  int nnz_H_union = 0;
  int nnz_AiTAi_lower = 0;  // entries of tril(A_i^T A_i)
  // the reference's choice between its sparse and its dense LDLT (interior_point.hpp:340-352, sqp.hpp:238-240,
  // newton.hpp:133-135): sparse iff nnz(H) + nnz(tril(A_i^T A_i)) + nnz(A_e) < 0.25 (n + m_e)^2
  bool reference_takes_dense(int nnz_Ae) const {
    return !(static_cast<double>(nnz_H_union) + nnz_AiTAi_lower + nnz_Ae < 0.25 * static_cast<double>(dim) * static_cast<double>(dim));
  }
};

KktPlan build_kkt_plan(const NlpStructure& s);

}  // namespace slpx
