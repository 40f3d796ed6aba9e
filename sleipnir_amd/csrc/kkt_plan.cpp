#include "kkt_plan.hpp"

#include "setup_timing.hpp"

#include <algorithm>

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
  SetupLap lap;
  k.n = s.n;
  k.m_e = s.m_e;
  k.m_i = s.m_i;
  k.dim = s.n + s.m_e;
  const int n = s.n;

  // every (col, row) that has a source, in the order the sources are found (H_f before H_c inside a column,
  // the products of a row of A_i in (a, b) order, rows ascending — the order the sums are taken in); the
  // entries themselves = the distinct keys in CSC order (col major, rows ascending)
  auto key_of = [](int32_t col, int32_t row) { return (static_cast<uint64_t>(static_cast<uint32_t>(col)) << 32) | static_cast<uint32_t>(row); };
  std::vector<uint64_t> dkey, pkey;
  std::vector<int32_t> dval, pav, pbv, prv;
  dkey.reserve(s.Hf.nnz() + s.Hc.nnz() + s.Ae.nnz());
  dval.reserve(dkey.capacity());
  // H = d_f H_f + H_c (problem.hpp:635); V already carries the d_f factor
  for (int c = 0; c < n; ++c) {
    for (int p = s.Hf.colptr[c]; p < s.Hf.colptr[c + 1]; ++p) {
      dkey.push_back(key_of(c, s.Hf.rowidx[p]));
      dval.push_back(s.off_Hf + p);
    }
    for (int p = s.Hc.colptr[c]; p < s.Hc.colptr[c + 1]; ++p) {
      dkey.push_back(key_of(c, s.Hc.rowidx[p]));
      dval.push_back(s.off_Hc + p);
    }
  }
  {
    std::vector<uint64_t> hk(dkey);
    std::sort(hk.begin(), hk.end());
    k.nnz_H_union = static_cast<int>(std::unique(hk.begin(), hk.end()) - hk.begin());
  }
  // tril(AᵢᵀΣAᵢ) (interior_point.hpp:432-433): entry (p,q), p>=q, sums over rows r
  csc_to_csr(s.Ai, s.off_Ai, k.ai_rowptr, k.ai_col, k.ai_src);
  csc_to_csr(s.Ae, s.off_Ae, k.ae_rowptr, k.ae_col, k.ae_src);
  for (int r = 0; r < s.m_i; ++r) {
    for (int a = k.ai_rowptr[r]; a < k.ai_rowptr[r + 1]; ++a)
      for (int b = k.ai_rowptr[r]; b < k.ai_rowptr[r + 1]; ++b) {
        int32_t row = k.ai_col[a], col = k.ai_col[b];
        if (row < col) continue;
        pkey.push_back(key_of(col, row));
        pav.push_back(k.ai_src[a]);  // (AᵢᵀΣ)(row, r) = Aᵢ(r,row) σ_r
        pbv.push_back(k.ai_src[b]);  // Aᵢ(r, col)
        prv.push_back(r);
      }
  }
  {
    std::vector<uint64_t> pk(pkey);
    std::sort(pk.begin(), pk.end());
    k.nnz_AiTAi_lower = static_cast<int>(std::unique(pk.begin(), pk.end()) - pk.begin());
  }
  // A_e block below (append_as_triplets.hpp:38-46, row offset n)
  for (int c = 0; c < n; ++c)
    for (int p = s.Ae.colptr[c]; p < s.Ae.colptr[c + 1]; ++p) {
      dkey.push_back(key_of(c, n + s.Ae.rowidx[p]));
      dval.push_back(s.off_Ae + p);
    }
  std::vector<uint64_t> keys;
  keys.reserve(dkey.size() + pkey.size() + k.dim);
  keys.insert(keys.end(), dkey.begin(), dkey.end());
  keys.insert(keys.end(), pkey.begin(), pkey.end());
  // forced diagonal (sparse_regularized_ldlt.hpp:67: lhs + regularization(0, 0))
  for (int d = 0; d < k.dim; ++d) keys.push_back(key_of(d, d));
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  const size_t nnz = keys.size();
  auto entry_of = [&](uint64_t key) { return static_cast<size_t>(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin()); };

  k.lhs.rows = k.lhs.cols = k.dim;
  k.lhs.colptr.assign(k.dim + 1, 0);
  k.lhs.rowidx.resize(nnz);
  for (size_t e = 0; e < nnz; ++e) {
    ++k.lhs.colptr[static_cast<int32_t>(keys[e] >> 32) + 1];
    k.lhs.rowidx[e] = static_cast<int32_t>(keys[e] & 0xffffffffu);
  }
  for (int c = 0; c < k.dim; ++c) k.lhs.colptr[c + 1] += k.lhs.colptr[c];
  // counting sort of the sources by entry: the order of discovery is kept inside an entry
  std::vector<uint32_t> dent(dkey.size()), pent(pkey.size());
  k.dptr.assign(nnz + 1, 0);
  k.pptr.assign(nnz + 1, 0);
  for (size_t i = 0; i < dkey.size(); ++i) ++k.dptr[(dent[i] = static_cast<uint32_t>(entry_of(dkey[i]))) + 1];
  for (size_t i = 0; i < pkey.size(); ++i) ++k.pptr[(pent[i] = static_cast<uint32_t>(entry_of(pkey[i]))) + 1];
  for (size_t e = 0; e < nnz; ++e) {
    k.dptr[e + 1] += k.dptr[e];
    k.pptr[e + 1] += k.pptr[e];
  }
  k.dsrc.resize(dkey.size());
  k.pa.resize(pkey.size());
  k.pb.resize(pkey.size());
  k.pr.resize(pkey.size());
  {
    std::vector<int32_t> dnext(k.dptr.begin(), k.dptr.end() - 1), pnext(k.pptr.begin(), k.pptr.end() - 1);
    for (size_t i = 0; i < dkey.size(); ++i) k.dsrc[dnext[dent[i]]++] = dval[i];
    for (size_t i = 0; i < pkey.size(); ++i) {
      const int32_t q = pnext[pent[i]]++;
      k.pa[q] = pav[i];
      k.pb[q] = pbv[i];
      k.pr[q] = prv[i];
    }
  }

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
  lap("  kkt plan");
  return k;
}

}  // namespace slpx
