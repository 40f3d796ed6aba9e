#include "kkt_plan.hpp"

#include <algorithm>
#include <map>

namespace slpx {

namespace {
struct Sources {
  std::vector<int32_t> direct;
  std::vector<int32_t> pa, pb, pr;
};
void csc_to_csr(const CscPattern& a, int voff, std::vector<int32_t>& rowptr,
                std::vector<int32_t>& col, std::vector<int32_t>& src) {
  rowptr.assign(a.rows + 1, 0);
  for (int32_t r : a.rowidx) ++rowptr[r + 1];
  for (int r = 0; r < a.rows; ++r) rowptr[r + 1] += rowptr[r];
  col.assign(a.nnz(), 0);
  src.assign(a.nnz(), 0);
  std::vector<int32_t> next(rowptr.begin(), rowptr.end() - 1);
  for (int c = 0; c < a.cols; ++c)
    for (int p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
      int q = next[a.rowidx[p]]++;
      col[q] = c;
      src[q] = voff + p;
    }
}
}  // namespace

KktPlan build_kkt_plan(const NlpStructure& s) {
  KktPlan k;
  k.n = s.n;
  k.m_e = s.m_e;
  k.m_i = s.m_i;
  k.dim = s.n + s.m_e;
  const int n = s.n;

  // (col, row) -> sources; std::map keeps CSC order (col major, rows ascending)
  std::map<std::pair<int32_t, int32_t>, Sources> ent;
  // H = d_f H_f + H_c (problem.hpp:635); V already carries the d_f factor
  for (int c = 0; c < n; ++c) {
    for (int p = s.Hf.colptr[c]; p < s.Hf.colptr[c + 1]; ++p)
      ent[{c, s.Hf.rowidx[p]}].direct.push_back(s.off_Hf + p);
    for (int p = s.Hc.colptr[c]; p < s.Hc.colptr[c + 1]; ++p)
      ent[{c, s.Hc.rowidx[p]}].direct.push_back(s.off_Hc + p);
  }
  k.nnz_H_union = static_cast<int>(ent.size());
  // tril(AᵢᵀΣAᵢ) (interior_point.hpp:432-433): entry (p,q), p>=q, sums over rows r
  csc_to_csr(s.Ai, s.off_Ai, k.ai_rowptr, k.ai_col, k.ai_src);
  csc_to_csr(s.Ae, s.off_Ae, k.ae_rowptr, k.ae_col, k.ae_src);
  for (int r = 0; r < s.m_i; ++r) {
    for (int a = k.ai_rowptr[r]; a < k.ai_rowptr[r + 1]; ++a)
      for (int b = k.ai_rowptr[r]; b < k.ai_rowptr[r + 1]; ++b) {
        int32_t row = k.ai_col[a], col = k.ai_col[b];
        if (row < col) continue;
        Sources& e = ent[{col, row}];
        e.pa.push_back(k.ai_src[a]);  // (AᵢᵀΣ)(row, r) = Aᵢ(r,row) σ_r
        e.pb.push_back(k.ai_src[b]);  // Aᵢ(r, col)
        e.pr.push_back(r);
      }
  }
  // A_e block below (append_as_triplets.hpp:38-46, row offset n)
  for (int c = 0; c < n; ++c)
    for (int p = s.Ae.colptr[c]; p < s.Ae.colptr[c + 1]; ++p)
      ent[{c, n + s.Ae.rowidx[p]}].direct.push_back(s.off_Ae + p);
  // forced diagonal (sparse_regularized_ldlt.hpp:67: lhs + regularization(0, 0))
  for (int d = 0; d < k.dim; ++d) ent[{d, d}];

  k.lhs.rows = k.lhs.cols = k.dim;
  k.lhs.colptr.assign(k.dim + 1, 0);
  k.dptr.push_back(0);
  k.pptr.push_back(0);
  for (auto& [key, src] : ent) {
    ++k.lhs.colptr[key.first + 1];
    k.lhs.rowidx.push_back(key.second);
    k.dsrc.insert(k.dsrc.end(), src.direct.begin(), src.direct.end());
    k.pa.insert(k.pa.end(), src.pa.begin(), src.pa.end());
    k.pb.insert(k.pb.end(), src.pb.begin(), src.pb.end());
    k.pr.insert(k.pr.end(), src.pr.begin(), src.pr.end());
    k.dptr.push_back(static_cast<int32_t>(k.dsrc.size()));
    k.pptr.push_back(static_cast<int32_t>(k.pa.size()));
  }
  for (int c = 0; c < k.dim; ++c) k.lhs.colptr[c + 1] += k.lhs.colptr[c];

  k.g_src.assign(n, -1);
  for (int c = 0; c < n; ++c)
    if (s.g_pat.colptr[c + 1] > s.g_pat.colptr[c]) k.g_src[c] = s.off_g + s.g_pat.colptr[c];

  // assemble = 12(h+a+i) + 16 m_i + 8 k ; rhs = 16n + 24m_e + 24m_i + 12(a+i)
  const int64_t h = s.Hf.nnz() + s.Hc.nnz(), a = s.Ae.nnz(), i = s.Ai.nnz(), kk = k.lhs.nnz();
  k.assemble_bytes = 12 * (h + a + i) + 16LL * s.m_i + 8 * kk;
  k.fast_src.resize(k.lhs.nnz());
  for (int e = 0; e < k.lhs.nnz(); ++e) {
    const int nd = k.dptr[e + 1] - k.dptr[e], np = k.pptr[e + 1] - k.pptr[e];
    k.fast_src[e] = (np == 0 && nd == 1) ? k.dsrc[k.dptr[e]] : ((np == 0 && nd == 0) ? -1 : -2);
  }
  k.rhs_bytes = 16LL * n + 24LL * s.m_e + 24LL * s.m_i + 12 * (a + i);
  return k;
}

}  // namespace slpx
