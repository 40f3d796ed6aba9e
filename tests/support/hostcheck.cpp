// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// Sequential host interpreter of the COMPILED device plans (tape tasks, KKT gather
// maps, LDLᵀ task/round schedule).  It walks exactly the data structures the HIP
// kernels walk, in the same task/level order, so the `-m "not gpu"` suite can
// validate the host-side compilers (tape_compiler, kkt_plan, ldlt_symbolic)
// against the oracle without a GPU.  The kernels themselves are validated on the
// GPU by the `-m gpu` parity tests.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../sleipnir_amd/csrc/capi_internal.hpp"
#include "../../sleipnir_amd/csrc/kkt_plan.hpp"
#include "../../sleipnir_amd/csrc/ldlt_symbolic.hpp"
#include "../../sleipnir_amd/csrc/nlp.hpp"
#include "../../sleipnir_amd/csrc/tape_jit.hpp"
#include "../../sleipnir_amd/csrc/tape_ops.h"

using namespace slpx;

struct hc_handle {
  NlpStructure s;
  KktPlan k;
  LdltPlan l;
  std::vector<double> scales, in_scale, V, lhs, rhs, Lx, D, contrib, scontrib, zv, xg, p, ps, pz;
  std::vector<double> z_factor;  // z left behind by the factorization (rhs carried as a row)
  std::vector<double> dense_A;   // the dense plan's factors (column-major, L below the diagonal, D on it)
  std::vector<int32_t> dense_trans;  // ... the pivoted one's transpositions
  // multifrontal plan (SLPX_LDLT_MF=1): the update slots between tasks, and every task's first
  // 64 KB of LDS as the factorization leaves it (the in-place backward solve reads U and 1/d there)
  std::vector<double> mf_contrib;
  std::vector<std::vector<double>> mf_lds;
  int stats[4] = {0, 0, 0, 0};
  double min_abs = 0.0;
};

namespace {

void run_tape(const TapeProgram& P, const std::vector<double>& in, const std::vector<double>& in_scale,
              const std::vector<double>& scales, std::vector<double>& V, bool reverse) {
  for (const TapeTask& t : P.tasks) {
    std::vector<double> val(t.n_leaf + t.n_node), part(2 * t.n_node), adj(t.n_slot);
    for (uint32_t i = 0; i < t.n_leaf; ++i) {
      uint32_t src = P.leaf_src[t.leaf_off + i];
      val[i] = (src & kLeafConstFlag) ? P.consts[src & ~kLeafConstFlag] : in[src] * in_scale[src];
    }
    const uint32_t* lvl = P.lvl_ptr.data() + t.lvl_off;
    const uint32_t* rec = P.node_rec.data() + 3 * static_cast<size_t>(t.node_off);
    for (uint32_t l = 0; l < t.n_lvl; ++l) {
      // emulate level parallelism: read everything of the level before writing
      std::vector<double> v(lvl[l + 1] - lvl[l]), dl(v.size()), dr(v.size());
      for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) {
        uint32_t r0 = rec[3 * i], a0 = rec[3 * i + 1], a1 = rec[3 * i + 2];
        op_forward(static_cast<Opcode>(r0 & 0xff), val[a0], val[a1], (r0 & 0x100) != 0,
                   (r0 & 0x200) != 0, v[i - lvl[l]], dl[i - lvl[l]], dr[i - lvl[l]]);
      }
      for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) {
        val[t.n_leaf + i] = v[i - lvl[l]];
        part[2 * i] = dl[i - lvl[l]];
        part[2 * i + 1] = dr[i - lvl[l]];
      }
    }
    for (uint32_t i = 0; i < t.n_vout; ++i) {
      uint32_t kx = t.vout_off + i;
      int32_t sc = P.vout_scale[kx];
      double v = val[P.vout_src[kx]];
      V[P.vout_dst[kx]] = sc >= 0 ? scales[sc] * v : v;
    }
    if (!reverse || t.n_slot == 0) continue;
    const uint32_t* slvl = P.slvl_ptr.data() + t.slvl_off;
    const uint32_t* eptr = P.slot_edge_ptr.data() + t.slot_off;
    const TapeEdge* edges = P.edges.data() + t.edge_off;
    for (uint32_t l = 0; l < t.n_slvl; ++l) {
      std::vector<double> acc(slvl[l + 1] - slvl[l]);
      for (uint32_t i = slvl[l]; i < slvl[l + 1]; ++i) {
        uint32_t eb = eptr[i], ee = eptr[i + 1];
        double a = eb == ee ? 1.0 : 0.0;
        for (uint32_t e = eb; e < ee; ++e) a += adj[edges[e].parent_slot] * part[edges[e].partial];
        acc[i - slvl[l]] = a;
      }
      for (uint32_t i = slvl[l]; i < slvl[l + 1]; ++i) adj[i] = acc[i - slvl[l]];
    }
    for (uint32_t i = 0; i < t.n_jout; ++i) {
      uint32_t kx = t.jout_off + i;
      int32_t sc = P.jout_scale[kx];
      double v = adj[P.jout_slot[kx]];
      V[P.jout_dst[kx]] = sc >= 0 ? scales[sc] * v : v;
    }
  }
}

}  // namespace

static void hc_build(hc_handle* h, slpx_problem* p, const int32_t* perm, int32_t perm_len,
                     int32_t task_entries, int32_t small_lds_bytes);

extern "C" {

hc_handle* hc_create(slpx_problem* p, const int32_t* perm, int32_t perm_len, int32_t task_entries,
                     int32_t small_lds_bytes) {
  auto* h = new hc_handle();
  try {
    hc_build(h, p, perm, perm_len, task_entries, small_lds_bytes);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "hc_create: %s\n", e.what());
    delete h;
    return nullptr;
  }
  return h;
}
}  // extern "C"

static void hc_build(hc_handle* h, slpx_problem* p, const int32_t* perm, int32_t perm_len,
                     int32_t task_entries, int32_t small_lds_bytes) {
  std::vector<NodeId> xs, ce, ci;
  for (auto& v : p->problem.decision_variables()) xs.push_back(v.expr);
  for (auto& v : p->problem.equality_constraints()) ce.push_back(v.expr);
  for (auto& v : p->problem.inequality_constraints()) ci.push_back(v.expr);
  NodeId f = p->problem.cost_function_type() == slp::ExpressionType::NONE ? kNull : p->problem.cost().expr;
  TapeCompileOptions topt;
  if (small_lds_bytes > 0) topt.small_lds_bytes = small_lds_bytes;
  h->s = build_nlp_structure(graph(), xs, f, ce, ci, topt);
  h->k = build_kkt_plan(h->s);
  std::vector<uint8_t> diag_has_source(h->k.dim, 0);
  for (int c = 0; c < h->k.dim; ++c)
    for (int q = h->k.lhs.colptr[c]; q < h->k.lhs.colptr[c + 1]; ++q)
      if (h->k.lhs.rowidx[q] == c)
        diag_has_source[c] = (h->k.dptr[q + 1] > h->k.dptr[q]) || (h->k.pptr[q + 1] > h->k.pptr[q]);
  std::vector<int32_t> up;
  if (perm && perm_len > 0) up.assign(perm, perm + perm_len);
  LdltOptions lopt;
  if (task_entries > 0) lopt.task_entries = task_entries;
  if (const char* env = std::getenv("SLPX_SUPERNODAL")) lopt.supernodal = env[0] != '0';
  if (const char* env = std::getenv("SLPX_SN_MIN_WIDTH")) lopt.min_supernode_width = std::atoi(env);
  if (const char* env = std::getenv("SLPX_SN_MAX_WIDTH")) lopt.max_supernode_width = static_cast<uint32_t>(std::atoi(env));
  if (const char* env = std::getenv("SLPX_SN_BALANCE")) lopt.balance_supernode_cuts = env[0] != '0';
  if (const char* env = std::getenv("SLPX_SN_DEEPEST")) {  // 0: off, 1: every task, 2: from round 1 up (default)
    lopt.chain_from_deepest_child = env[0] != '0';
    lopt.chain_from_deepest_min_round = env[0] == '1' ? 0 : 1;
  }
  if (const char* env = std::getenv("SLPX_HOSTCHECK_SN_MAX_WIDTH")) lopt.max_supernode_width = static_cast<uint32_t>(std::atoi(env));
  if (const char* env = std::getenv("SLPX_LDLT_MF"))
    if (env[0] != '0') {
      lopt.multifrontal = true;
      if (std::getenv("SLPX_SN_MIN_WIDTH") == nullptr) lopt.min_supernode_width = 2;
      lopt.relax_zeros = 8;
      if (std::getenv("SLPX_SN_BALANCE") == nullptr) lopt.balance_supernode_cuts = true;
      if (std::getenv("SLPX_SN_DEEPEST") == nullptr) {
        lopt.chain_from_deepest_child = true;
        lopt.chain_from_deepest_min_round = 0;
      }
    }
  if (const char* env = std::getenv("SLPX_RELAX_ZEROS")) lopt.relax_zeros = std::atoi(env);
  if (const char* env = std::getenv("SLPX_MFMA_MIN_ENTRIES")) lopt.mfma_min_entries = static_cast<uint32_t>(std::atoi(env));
  // (the product's rule, newton.cpp: plan_or_dense)
  const char* dense_env = std::getenv("SLPX_DENSE");
  if (h->k.reference_takes_dense(h->s.Ae.nnz()) && up.empty() && (dense_env == nullptr || dense_env[0] != '0') && h->k.dim <= 2048) {
    h->l = build_dense_ldlt_plan(h->k.lhs, h->s.n);
    h->l.dense_pivoted = dense_env == nullptr || dense_env[0] != '1';
  } else if (dense_env != nullptr && dense_env[0] == '1') {
    h->l = build_dense_ldlt_plan(h->k.lhs, h->s.n);
  } else {
    try {
      h->l = build_ldlt_plan(h->k.lhs, h->s.n, lopt, up.empty() ? nullptr : &up, &diag_has_source);
    } catch (const std::runtime_error& e) {
      if (!ldlt_plan_error_is_too_big(e)) throw;
      h->l = build_dense_ldlt_plan(h->k.lhs, h->s.n);
    }
  }
  h->scales.assign(h->s.n_scales(), 1.0);
  h->in_scale.assign(h->s.n_inputs(), 1.0);
  h->V = h->s.V_static_raw;
  h->lhs.assign(h->k.lhs.nnz(), 0.0);
  h->rhs.assign(h->k.dim, 0.0);
  h->Lx.assign(std::max<int64_t>(1, h->l.nnzL), 0.0);
  h->D.assign(h->l.n, 0.0);
  h->contrib.assign(std::max<uint32_t>(1, h->l.n_contrib), 0.0);
  h->mf_contrib.assign(std::max<uint32_t>(1, h->l.mf_n_contrib), 0.0);
  h->mf_lds.assign(h->l.mf ? h->l.tasks.size() : 0, {});
  h->scontrib.assign(std::max<uint32_t>(1, h->l.n_scontrib), 0.0);
  h->zv.assign(h->l.n, 0.0);
  h->z_factor.assign(h->l.n, 0.0);
  h->xg.assign(h->l.n, 0.0);
  h->p.assign(h->k.dim, 0.0);
  h->ps.assign(std::max(1, h->s.m_i), 0.0);
  h->pz.assign(std::max(1, h->s.m_i), 0.0);
  // lists the structural families of the tape (stderr); no device needed for that part
  if (std::getenv("SLPX_TAPE_JIT_VERBOSE")) (void)build_tape_templates(h->s.full);
}

// ---------------------------------------------------------------------------
// Where a multifrontal step loses its digits: the plan run twice, in double and in long
// double (64-bit significand), front by front in the kernel's order, and the two compared at every level: the
// finished pivot columns and the update block of every front, then the x of every front of the backward solve.
// out rows: {phase (0 factorization, 1 backward solve), round, level, values compared,
//            max |d - q| / max|q| over the level's fronts (normwise, per front), median and max of |d - q| / |q|}
// ---------------------------------------------------------------------------
namespace {
struct MfLogEntry {
  uint32_t phase, round, level, front;
  long double value;
};
template <class T>
void mf_run_logged(const hc_handle* h, double delta, double gamma, std::vector<MfLogEntry>& log) {
  const LdltPlan& L = h->l;
  std::vector<T> contrib(std::max<uint32_t>(1, L.mf_n_contrib), T(0)), xg(L.n, T(0));
  std::vector<std::vector<T>> lds_all(L.tasks.size());
  uint32_t front_id = 0;
  for (int r = 0; r < L.n_rounds; ++r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      const LdltMfTask& M = L.mf_tasks[ti];
      const uint32_t off_arena = t.n_ent, off_invd = off_arena + M.arena, off_x = off_invd + t.n_col;
      std::vector<T>& lds = lds_all[ti];
      lds.assign(off_x + t.n_col + M.n_anc + 1, T(0));
      auto at = [&](uint16_t byte_off) -> T& { return lds[byte_off / 8u]; };
      const uint32_t* cptr = L.mf_contrib_ptr.data() + M.contrib_ptr_off;
      const uint32_t* cidx = L.mf_contrib_idx.data() + M.contrib_off;
      const uint16_t* cent = L.mf_cent.data() + M.cent_off;
      for (uint32_t i = 0; i < t.n_ent; ++i) {
        const uint32_t e = t.ent_off + i;
        const int32_t src = L.ent_src[e];
        const uint8_t fl = L.ent_flags[e];
        T acc = src >= 0 ? T((fl & 4) ? h->rhs[src] : h->lhs[src]) : T(0);
        if (fl & 1) acc += (fl & 2) ? T(-gamma) : T(delta);
        lds[i] = acc;
      }
      for (uint32_t j = 0; j < M.n_cent; ++j)
        for (uint32_t c = cptr[j]; c < cptr[j + 1]; ++c) lds[cent[j]] -= contrib[cidx[c]];
      const uint32_t* lvl = L.mf_lvl_ptr.data() + t.lvl_off;
      const uint16_t* tab0 = L.mf_tab.data() + M.tab_off;
      const uint32_t* ext = L.mf_ext.data() + M.ext_off;
      for (uint32_t l = 0; l < t.n_lvl; ++l)
        for (uint32_t q = lvl[l]; q < lvl[l + 1]; ++q, ++front_id) {
          const LdltFront& F = L.mf_fronts[M.front_off + q];
          const uint32_t w = F.w, nr = F.nr, nch = F.nch;
          const uint16_t* piv = tab0 + F.tab;
          const uint16_t* upd = piv + static_cast<size_t>(nr) * (1 + nch) * w;
          std::vector<T> a(static_cast<size_t>(nr) * w, T(0)), inv(w);
          for (uint32_t row = 0; row < nr; ++row)
            for (uint32_t c = 0; c < w && c <= row; ++c) {
              T v = 0;
              for (uint32_t k = 0; k <= nch; ++k) v += at(piv[(static_cast<size_t>(row) * (1 + nch) + k) * w + c]);
              a[static_cast<size_t>(row) * w + c] = v;
            }
          for (uint32_t c = 0; c < w; ++c) {
            inv[c] = T(1) / a[static_cast<size_t>(c) * w + c];
            for (uint32_t row = c + 1; row < nr; ++row) {
              const T lc = a[static_cast<size_t>(row) * w + c] * inv[c];
              for (uint32_t j = c + 1; j < w && j <= row; ++j) a[static_cast<size_t>(row) * w + j] -= lc * a[static_cast<size_t>(j) * w + c];
            }
          }
          for (uint32_t row = 0; row < nr; ++row)
            for (uint32_t c = 0; c < w && c <= row; ++c) {
              at(piv[(static_cast<size_t>(row) * (1 + nch)) * w + c]) = a[static_cast<size_t>(row) * w + c];
              log.push_back({0u, static_cast<uint32_t>(r), l, front_id, static_cast<long double>(a[static_cast<size_t>(row) * w + c])});
            }
          for (uint32_t c = 0; c < w; ++c) lds[off_invd + F.col0 + c] = inv[c];
          for (uint32_t e = 0; e < F.n_s; ++e) {
            const uint16_t* row = upd + static_cast<size_t>(e) * (3 + nch);
            T v = 0;
            for (uint32_t k = 0; k < nch; ++k) v += at(row[3 + k]);
            for (uint32_t c = 0; c < w; ++c) {
              const uint32_t coff = c * nr - (c * (c - 1)) / 2 - c;
              v -= (lds[row[1] / 8u + coff] * inv[c]) * lds[row[2] / 8u + coff];
            }
            if (F.flags & 1) contrib[ext[F.ext + row[0]]] = -v;
            else at(row[0]) = v;
            log.push_back({0u, static_cast<uint32_t>(r), l, front_id, static_cast<long double>(v)});
          }
        }
    }
  for (int r = L.n_rounds - 1; r >= 0; --r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      const LdltMfTask& M = L.mf_tasks[ti];
      const uint32_t off_invd = t.n_ent + M.arena, off_x = off_invd + t.n_col;
      std::vector<T>& lds = lds_all[ti];
      for (uint32_t a = 0; a < M.n_anc; ++a) lds[off_x + t.n_col + a] = xg[L.mf_anc[M.anc_off + a]];
      const uint32_t* lvl = L.mf_lvl_ptr.data() + t.lvl_off;
      const uint16_t* tab0 = L.mf_tab.data() + M.tab_off;
      for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l)
        for (uint32_t q = lvl[l]; q < lvl[l + 1]; ++q) {
          const LdltFront& F = L.mf_fronts[M.front_off + q];
          const uint32_t w = F.w, nr = F.nr, nch = F.nch, rr = nr - w - 1;
          const uint16_t* xr = tab0 + F.tab + static_cast<size_t>(nr) * (1 + nch) * w + static_cast<size_t>(F.n_s) * (3 + nch);
          auto U = [&](uint32_t row, uint32_t c) { return lds[F.base0 + c * nr - (c * (c - 1)) / 2 + (row - c)]; };
          for (int c = static_cast<int>(w) - 1; c >= 0; --c) {
            T dot = 0;
            for (uint32_t a = 0; a < rr; ++a) dot += U(w + a, c) * lds[xr[a] / 8u];
            for (uint32_t k = c + 1; k < w; ++k) dot += U(k, c) * lds[off_x + F.col0 + k];
            lds[off_x + F.col0 + c] = (U(nr - 1, c) - dot) * lds[off_invd + F.col0 + c];
            log.push_back({1u, static_cast<uint32_t>(r), static_cast<uint32_t>(l), ti * 65536u + q, static_cast<long double>(lds[off_x + F.col0 + c])});
          }
        }
      for (uint32_t i = 0; i < t.n_col; ++i) xg[L.col_perm[t.col_off + i]] = lds[off_x + i];
    }
}
}  // namespace

extern "C" int32_t hc_mf_level_errors(hc_handle* h, double delta, double gamma, double* out, int32_t cap_rows) {
  if (!h->l.mf) return -1;
  std::vector<MfLogEntry> ld, dd;
  mf_run_logged<long double>(h, delta, gamma, ld);
  mf_run_logged<double>(h, delta, gamma, dd);
  if (ld.size() != dd.size()) return -2;
  // fronts are logged task by task: gather every task's fronts of a (phase, round, level)
  struct Acc {
    std::vector<double> comp;
    double worst_norm = 0.0;
    size_t values = 0;
  };
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, Acc> acc;
  std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> order;  // in the order of first appearance
  size_t a = 0;
  while (a < ld.size()) {  // one front
    size_t b = a;
    long double big = 0, diff = 0;
    const auto key = std::make_tuple(ld[a].phase, ld[a].round, ld[a].level);
    if (acc.find(key) == acc.end()) order.push_back(key);
    Acc& A = acc[key];
    while (b < ld.size() && ld[b].front == ld[a].front && ld[b].phase == ld[a].phase) {
      big = std::max(big, std::fabs(ld[b].value));
      const long double d = std::fabs(dd[b].value - ld[b].value);
      diff = std::max(diff, d);
      if (ld[b].value != 0) A.comp.push_back(static_cast<double>(d / std::fabs(ld[b].value)));
      ++b;
    }
    A.values += b - a;
    if (big > 0) A.worst_norm = std::max(A.worst_norm, static_cast<double>(diff / big));
    a = b;
  }
  int32_t rows = 0;
  for (const auto& key : order) {
    Acc& A = acc[key];
    std::sort(A.comp.begin(), A.comp.end());
    if (rows < cap_rows) {
      double* o = out + 7 * static_cast<size_t>(rows);
      o[0] = std::get<0>(key);
      o[1] = std::get<1>(key);
      o[2] = std::get<2>(key);
      o[3] = static_cast<double>(A.values);
      o[4] = A.worst_norm;
      o[5] = A.comp.empty() ? 0.0 : A.comp[A.comp.size() / 2];
      o[6] = A.comp.empty() ? 0.0 : A.comp.back();
    }
    ++rows;
  }
  return rows;
}

// FNV-1a over every array of the compiled plans: setup refactorings (threads, flat arrays) are checked
// to leave the plans the same to the byte (tests/test_plans_cpu.py::test_plans_do_not_depend_on_the_thread_count).
namespace {
struct PlanHash {
  uint64_t h = 1469598103934665603ull;
  void bytes(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  }
  template <class T> void vec(const std::vector<T>& v) {
    const uint64_t n = v.size();
    bytes(&n, sizeof n);
    if (!v.empty()) bytes(v.data(), v.size() * sizeof(T));
  }
  template <class T> void pod(const T& v) { bytes(&v, sizeof v); }
};
uint64_t hash_tape(const TapeProgram& P) {
  PlanHash H;
  H.pod(P.n_inputs); H.pod(P.n_outputs);
  H.vec(P.tasks); H.vec(P.small_tasks); H.vec(P.large_tasks); H.vec(P.global_tasks); H.vec(P.leaf_src); H.vec(P.consts);
  for (auto& pr : P.params) { H.pod(pr.first); H.pod(pr.second); }
  H.pod(P.shared_tasks);
  H.vec(P.node_rec); H.vec(P.lvl_ptr); H.vec(P.slot_edge_ptr); H.vec(P.slvl_ptr); H.vec(P.edges);
  H.vec(P.node_rec16); H.vec(P.slot_edge_ptr16); H.vec(P.edges16);
  H.vec(P.vout_src); H.vec(P.vout_dst); H.vec(P.vout_scale); H.vec(P.jout_slot); H.vec(P.jout_dst); H.vec(P.jout_scale);
  H.pod(P.small_lds_bytes); H.pod(P.large_lds_bytes); H.pod(P.global_scratch_doubles);
  H.pod(P.total_nodes); H.pod(P.total_slots); H.pod(P.total_edges); H.pod(P.total_leaves); H.pod(P.max_levels); H.pod(P.max_slot_levels);
  return H.h;
}
void hash_csc(PlanHash& H, const CscPattern& c) { H.pod(c.rows); H.pod(c.cols); H.vec(c.colptr); H.vec(c.rowidx); }
}  // namespace

extern "C" int32_t hc_is_dense(hc_handle* h) { return h->l.dense ? (h->l.dense_pivoted ? 2 : 1) : 0; }

extern "C" void hc_plan_hash(hc_handle* h, uint64_t* out) {
  {
    PlanHash H;
    const NlpStructure& s = h->s;
    H.pod(s.n); H.pod(s.m_e); H.pod(s.m_i); H.pod(s.nV); H.pod(s.off_g); H.pod(s.off_Ae); H.pod(s.off_Ai); H.pod(s.off_Hf); H.pod(s.off_Hc);
    hash_csc(H, s.g_pat); hash_csc(H, s.Ae); hash_csc(H, s.Ai); hash_csc(H, s.Hf); hash_csc(H, s.Hc);
    H.vec(s.V_static_raw); H.vec(s.V_scale_idx); H.vec(s.V_is_static);
    for (auto& r : s.reduces) H.pod(r);
    out[0] = H.h;
  }
  out[1] = hash_tape(h->s.full);
  out[2] = hash_tape(h->s.values);
  {
    PlanHash H;
    const KktPlan& k = h->k;
    hash_csc(H, k.lhs);
    H.vec(k.dptr); H.vec(k.dsrc); H.vec(k.pptr); H.vec(k.pa); H.vec(k.pb); H.vec(k.pr); H.vec(k.fast_src); H.vec(k.g_src);
    H.vec(k.ai_rowptr); H.vec(k.ai_col); H.vec(k.ai_src); H.vec(k.ae_rowptr); H.vec(k.ae_col); H.vec(k.ae_src);
    H.pod(k.assemble_bytes); H.pod(k.rhs_bytes); H.pod(k.nnz_H_union);
    out[3] = H.h;
  }
  {
    PlanHash H;
    const LdltPlan& l = h->l;
    H.pod(l.n); H.pod(l.n_dec); H.vec(l.perm); H.vec(l.iperm); H.vec(l.parent); H.vec(l.Lp); H.vec(l.Li); H.pod(l.nnzL);
    H.pod(l.etree_height); H.pod(l.n_rounds); H.pod(l.structurally_singular_unregularized);
    H.vec(l.tasks); H.vec(l.round_ptr); H.pod(l.max_lds_doubles); H.pod(l.max_solve_lds_doubles); H.pod(l.factor_lds_bytes); H.pod(l.solve_lds_bytes);
    H.vec(l.ent_src); H.vec(l.ent_flags); H.vec(l.ent_col); H.vec(l.ent_out); H.vec(l.ent_pair_ptr); H.vec(l.ent_contrib_ptr);
    H.vec(l.contrib_idx); H.vec(l.ext_dst); H.vec(l.lvl_ptr); H.vec(l.pairs); H.pod(l.n_contrib);
    H.vec(l.col_perm); H.vec(l.col_lvl_ptr); H.vec(l.fwd_ptr); H.vec(l.fwd_contrib_ptr); H.vec(l.scontrib_idx); H.vec(l.fwd_items);
    H.vec(l.sext_ptr); H.vec(l.sext_dst); H.vec(l.sext_items); H.vec(l.bwd_ptr); H.vec(l.bwd_items); H.pod(l.n_scontrib);
    H.vec(l.sn_desc); H.vec(l.sn_lvl_ptr); H.vec(l.col_sn); H.pod(l.n_supernodes); H.pod(l.widest_supernode); H.pod(l.critical_levels);
    H.vec(l.sn_width_hist);
    out[4] = H.h;
    PlanHash M;
    M.pod(l.mf); M.vec(l.mf_tasks); M.vec(l.mf_fronts); M.vec(l.mf_lvl_ptr); M.vec(l.mf_tab); M.vec(l.mf_ext); M.vec(l.mf_contrib_ptr);
    M.vec(l.mf_contrib_idx); M.vec(l.mf_cent); M.vec(l.mf_anc); M.pod(l.mf_n_contrib); M.pod(l.mf_max_nch); M.pod(l.mf_max_front_rows); M.pod(l.mf_n_mfma);
    M.pod(l.factor_bytes); M.pod(l.solve_bytes); M.pod(l.flops);
    out[5] = M.h;
  }
}

extern "C" {
void hc_destroy(hc_handle* h) { delete h; }

// Generates the run-time tape kernel of the full program and compiles it with hipRTC for
// gfx950 (no device needed).  Returns the number of bodies compiled (0: the program has no
// family worth one), -1 when hipRTC rejects the source (log on stderr).
int32_t hc_tape_jit_compiles(hc_handle* h) {
  TapeJitOptions opt;
  opt.compile_without_device = true;
  const TapeJitResult r = build_tape_templates(h->s.full, opt);
  if (r.compiled_without_device < 0) std::fprintf(stderr, "hc_tape_jit_compiles: %s\n", r.log.c_str());
  return r.compiled_without_device;
}

// Number of distinct task STRUCTURES of the LDLᵀ plan (everything but the global indices:
// sizes, levels, local columns, flags, pair lists); out[r] = distinct structures of round r.
int32_t hc_ldlt_families(hc_handle* h, int32_t* out, int32_t cap) {
  const LdltPlan& L = h->l;
  std::map<std::string, int> all;
  for (int r = 0; r < L.n_rounds; ++r) {
    std::map<std::string, int> fam;
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      std::string key;
      auto put = [&](const void* p, size_t n) { key.append(static_cast<const char*>(p), n); };
      const uint32_t head[5] = {t.n_ent, t.n_col, t.n_ext, t.n_lvl, t.n_pairs};
      put(head, sizeof(head));
      put(L.lvl_ptr.data() + t.lvl_off, 4 * (t.n_lvl + 1));
      put(L.ent_col.data() + t.ent_off, 2 * t.n_ent);
      put(L.ent_flags.data() + t.ent_off, t.n_ent);
      put(L.ent_pair_ptr.data() + t.pair_ptr_off, 4 * (t.n_ent + t.n_ext + 1));
      put(L.ent_contrib_ptr.data() + t.contrib_ptr_off, 4 * (t.n_ent + 1));
      put(L.pairs.data() + t.pair_off, sizeof(LdltPair) * t.n_pairs);
      ++fam[key];
      ++all[key];
    }
    if (r < cap) out[r] = static_cast<int32_t>(fam.size());
    if (std::getenv("SLPX_TAPE_JIT_VERBOSE"))
      for (auto& [k, c] : fam) {
        uint32_t hd[5];
        std::memcpy(hd, k.data(), sizeof(hd));
        std::fprintf(stderr, "ldlt round %d: %d x (ent %u col %u ext %u lvl %u pairs %u)\n", r, c, hd[0], hd[1],
                     hd[2], hd[3], hd[4]);
      }
  }
  return static_cast<int32_t>(all.size());
}

// out[0..] = n, m_e, m_i, nV, nnz_lhs, nnzL, rounds, tasks, etree height, pairs,
//            tape tasks, nodes, slots, edges, levels, slot levels, struct singular,
//            off_g, off_Ae, off_Ai, off_Hf, off_Hc, global tape tasks, large tape tasks
void hc_info(hc_handle* h, int64_t* out) {
  int i = 0;
  out[i++] = h->s.n;
  out[i++] = h->s.m_e;
  out[i++] = h->s.m_i;
  out[i++] = h->s.nV;
  out[i++] = h->k.lhs.nnz();
  out[i++] = h->l.nnzL;
  out[i++] = h->l.n_rounds;
  out[i++] = static_cast<int64_t>(h->l.tasks.size());
  out[i++] = h->l.etree_height;
  out[i++] = h->l.flops / 2;
  out[i++] = static_cast<int64_t>(h->s.full.tasks.size());
  out[i++] = static_cast<int64_t>(h->s.full.total_nodes);
  out[i++] = static_cast<int64_t>(h->s.full.total_slots);
  out[i++] = static_cast<int64_t>(h->s.full.total_edges);
  out[i++] = h->s.full.max_levels;
  out[i++] = h->s.full.max_slot_levels;
  out[i++] = h->l.structurally_singular_unregularized ? 1 : 0;
  out[i++] = h->s.off_g;
  out[i++] = h->s.off_Ae;
  out[i++] = h->s.off_Ai;
  out[i++] = h->s.off_Hf;
  out[i++] = h->s.off_Hc;
  out[i++] = static_cast<int64_t>(h->s.full.global_tasks.size());
  out[i++] = static_cast<int64_t>(h->s.full.large_tasks.size());
  out[i++] = h->s.full.shared_tasks;
  out[i++] = static_cast<int64_t>(2 * (h->s.full.node_rec16.size() + h->s.full.edges16.size() +
                                       h->s.full.slot_edge_ptr16.size()));
}

int32_t hc_pattern(hc_handle* h, int which, int32_t* colptr, int32_t* rowidx) {
  const CscPattern* pat = nullptr;
  switch (which) {
    case 0: pat = &h->s.g_pat; break;
    case 1: pat = &h->s.Ae; break;
    case 2: pat = &h->s.Ai; break;
    case 3: pat = &h->s.Hf; break;
    case 4: pat = &h->s.Hc; break;
    case 5: pat = &h->k.lhs; break;
    default: return -1;
  }
  if (colptr) std::copy(pat->colptr.begin(), pat->colptr.end(), colptr);
  if (rowidx) std::copy(pat->rowidx.begin(), pat->rowidx.end(), rowidx);
  return pat->nnz();
}
void hc_perm(hc_handle* h, int32_t* perm) { std::copy(h->l.perm.begin(), h->l.perm.end(), perm); }

void hc_set_scaling(hc_handle* h, const double* scales) {
  const auto& s = h->s;
  h->scales.assign(scales, scales + s.n_scales());
  for (int j = 0; j < s.m_e; ++j) h->in_scale[s.n + j] = scales[1 + j];
  for (int j = 0; j < s.m_i; ++j) h->in_scale[s.n + s.m_e + j] = scales[1 + s.m_e + j];
  for (int kx = 0; kx < s.nV; ++kx) {
    int32_t sc = s.V_scale_idx[kx];
    if (s.V_is_static[kx]) h->V[kx] = sc >= 0 ? scales[sc] * s.V_static_raw[kx] : s.V_static_raw[kx];
  }
}

void hc_sweep(hc_handle* h, const double* x, const double* y, const double* z, int full, double* V_out) {
  const auto& s = h->s;
  std::vector<double> in(s.n_inputs(), 0.0);
  std::copy(x, x + s.n, in.begin());
  if (y) std::copy(y, y + s.m_e, in.begin() + s.n);
  if (z) std::copy(z, z + s.m_i, in.begin() + s.n + s.m_e);
  run_tape(full ? s.full : s.values, in, h->in_scale, h->scales, h->V, full != 0);
  // separable sums: V[dst] = scale * sum of the partials (nlp.cpp, tape_reduce_kernel)
  for (const auto& r : s.reduces) {
    double acc = 0.0;
    for (int k = 0; k < r.count; ++k) acc += h->V[r.src_off + k];
    h->V[r.dst] = (r.scale_idx >= 0 ? h->scales[r.scale_idx] : 1.0) * acc;
  }
  if (V_out) std::copy(h->V.begin(), h->V.end(), V_out);
}

void hc_assemble(hc_handle* h, const double* s, const double* z, double* lhs_out) {
  const auto& K = h->k;
  for (int kx = 0; kx < K.lhs.nnz(); ++kx) {
    double direct = 0.0;
    for (int d = K.dptr[kx]; d < K.dptr[kx + 1]; ++d) direct += h->V[K.dsrc[d]];
    double prod = 0.0;
    for (int q = K.pptr[kx]; q < K.pptr[kx + 1]; ++q) {
      int r = K.pr[q];
      double sigma = (1.0 / s[r]) * z[r];
      prod += (h->V[K.pa[q]] * sigma) * h->V[K.pb[q]];
    }
    h->lhs[kx] = direct + prod;
  }
  if (lhs_out) std::copy(h->lhs.begin(), h->lhs.end(), lhs_out);
}
void hc_set_lhs(hc_handle* h, const double* lhs) { std::copy(lhs, lhs + h->k.lhs.nnz(), h->lhs.begin()); }

void hc_rhs(hc_handle* h, const double* s, const double* y, const double* z, double mu, double* rhs_out) {
  const auto& K = h->k;
  const auto& st = h->s;
  const double* V = h->V.data();
  for (int j = 0; j < K.dim; ++j) {
    if (j >= K.n) {
      h->rhs[j] = -V[st.off_ce + j - K.n];
      continue;
    }
    int gs = K.g_src[j];
    double aey = 0.0;
    for (int q = st.Ae.colptr[j]; q < st.Ae.colptr[j + 1]; ++q) aey += V[st.off_Ae + q] * y[st.Ae.rowidx[q]];
    double ait = 0.0;
    for (int q = st.Ai.colptr[j]; q < st.Ai.colptr[j + 1]; ++q) {
      int r = st.Ai.rowidx[q];
      double sinv = 1.0 / s[r], sigma = sinv * z[r];
      ait += V[st.off_Ai + q] * (-sigma * V[st.off_ci + r] + mu * sinv + z[r]);
    }
    h->rhs[j] = -(gs >= 0 ? V[gs] : 0.0) + aey + ait;
  }
  if (rhs_out) std::copy(h->rhs.begin(), h->rhs.end(), rhs_out);
}
void hc_set_rhs(hc_handle* h, const double* rhs) { std::copy(rhs, rhs + h->k.dim, h->rhs.begin()); }

// stats_out: n_pos, n_neg, n_zero, n_bad, min|D|
// The multifrontal plan (ldlt_symbolic.hpp: LdltFront) interpreted the way ldlt_mf_kernels.h runs
// it: every table word is a byte offset into the task's first 64 KB of LDS.
static void hc_factor_mf(hc_handle* h, double delta, double gamma) {
  const LdltPlan& L = h->l;
  for (int r = 0; r < L.n_rounds; ++r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      const LdltMfTask& M = L.mf_tasks[ti];
      const uint32_t off_arena = t.n_ent, off_invd = off_arena + M.arena, off_x = off_invd + t.n_col;
      std::vector<double>& lds = h->mf_lds[ti];
      lds.assign(off_x + t.n_col + M.n_anc + 1, 0.0);
      auto at = [&](uint16_t byte_off) -> double& { return lds[byte_off / 8u]; };
      const uint32_t* cptr = L.mf_contrib_ptr.data() + M.contrib_ptr_off;
      const uint32_t* cidx = L.mf_contrib_idx.data() + M.contrib_off;
      const uint16_t* cent = L.mf_cent.data() + M.cent_off;
      for (uint32_t i = 0; i < t.n_ent; ++i) {
        const uint32_t e = t.ent_off + i;
        const int32_t src = L.ent_src[e];
        const uint8_t fl = L.ent_flags[e];
        double acc = src >= 0 ? ((fl & 4) ? h->rhs[src] : h->lhs[src]) : 0.0;
        if (fl & 1) acc += (fl & 2) ? -gamma : delta;
        lds[i] = acc;
      }
      for (uint32_t j = 0; j < M.n_cent; ++j)
        for (uint32_t c = cptr[j]; c < cptr[j + 1]; ++c) lds[cent[j]] -= h->mf_contrib[cidx[c]];
      const uint32_t* lvl = L.mf_lvl_ptr.data() + t.lvl_off;
      const uint16_t* tab0 = L.mf_tab.data() + M.tab_off;
      const uint32_t* ext = L.mf_ext.data() + M.ext_off;
      for (uint32_t l = 0; l < t.n_lvl; ++l) {
        // the fronts of a level run concurrently on the device: they may only read what earlier levels wrote
        for (uint32_t q = lvl[l]; q < lvl[l + 1]; ++q) {
          const LdltFront& F = L.mf_fronts[M.front_off + q];
          const uint32_t w = F.w, nr = F.nr, nch = F.nch, rr = nr - w - 1;
          const uint16_t* piv = tab0 + F.tab;
          const uint16_t* upd = piv + static_cast<size_t>(nr) * (1 + nch) * w;
          std::vector<double> a(static_cast<size_t>(nr) * w, 0.0), inv(w);
          for (uint32_t row = 0; row < nr; ++row)
            for (uint32_t c = 0; c < w && c <= row; ++c) {
              double v = 0.0;
              for (uint32_t k = 0; k <= nch; ++k) v += at(piv[(static_cast<size_t>(row) * (1 + nch) + k) * w + c]);
              a[static_cast<size_t>(row) * w + c] = v;
            }
          for (uint32_t c = 0; c < w; ++c) {
            inv[c] = 1.0 / a[static_cast<size_t>(c) * w + c];
            for (uint32_t row = c + 1; row < nr; ++row) {
              const double lc = a[static_cast<size_t>(row) * w + c] * inv[c];
              for (uint32_t j = c + 1; j < w && j <= row; ++j) a[static_cast<size_t>(row) * w + j] -= lc * a[static_cast<size_t>(j) * w + c];
            }
          }
          for (uint32_t row = 0; row < nr; ++row)
            for (uint32_t c = 0; c < w && c <= row; ++c) at(piv[(static_cast<size_t>(row) * (1 + nch)) * w + c]) = a[static_cast<size_t>(row) * w + c];
          for (uint32_t c = 0; c < w; ++c) lds[off_invd + F.col0 + c] = inv[c];
          for (uint32_t e = 0; e < F.n_s; ++e) {
            const uint16_t* row = upd + static_cast<size_t>(e) * (3 + nch);
            double v = 0.0;
            for (uint32_t k = 0; k < nch; ++k) v += at(row[3 + k]);
            for (uint32_t c = 0; c < w; ++c) {
              const uint32_t coff = c * nr - (c * (c - 1)) / 2 - c;  // column c against column 0, same row
              v -= (lds[row[1] / 8u + coff] * inv[c]) * lds[row[2] / 8u + coff];
            }
            if (F.flags & 1) h->mf_contrib[ext[F.ext + row[0]]] = -v;
            else at(row[0]) = v;
          }
          (void)rr;
        }
      }
      for (uint32_t i = 0; i < t.n_ent; ++i) {
        const uint32_t e = t.ent_off + i;
        const double u = lds[i];
        if (L.ent_flags[e] & 1) {
          h->D[L.ent_out[e]] = u;
          const double eps = 2.220446049250313e-16;
          if (u > eps) ++h->stats[0];
          else if (u < -eps) ++h->stats[1];
          else ++h->stats[2];
          if (u == 0.0 || !std::isfinite(u)) ++h->stats[3];
          else h->min_abs = std::min(h->min_abs, std::fabs(u));
        } else if (L.ent_flags[e] & 4) {
          h->z_factor[L.ent_out[e]] = u * lds[off_invd + L.ent_col[e]];
        } else {
          h->Lx[L.ent_out[e]] = u * lds[off_invd + L.ent_col[e]];
        }
      }
    }
}

// backward substitution on what hc_factor_mf left in every task's LDS image (the fused step's solve)
static void hc_backward_mf(hc_handle* h, double* p_out) {
  const LdltPlan& L = h->l;
  for (int r = L.n_rounds - 1; r >= 0; --r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      const LdltMfTask& M = L.mf_tasks[ti];
      const uint32_t off_invd = t.n_ent + M.arena, off_x = off_invd + t.n_col;
      std::vector<double>& lds = h->mf_lds[ti];
      for (uint32_t a = 0; a < M.n_anc; ++a) lds[off_x + t.n_col + a] = h->xg[L.mf_anc[M.anc_off + a]];
      const uint32_t* lvl = L.mf_lvl_ptr.data() + t.lvl_off;
      const uint16_t* tab0 = L.mf_tab.data() + M.tab_off;
      for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l)
        for (uint32_t q = lvl[l]; q < lvl[l + 1]; ++q) {
          const LdltFront& F = L.mf_fronts[M.front_off + q];
          const uint32_t w = F.w, nr = F.nr, nch = F.nch, rr = nr - w - 1;
          const uint16_t* xr = tab0 + F.tab + static_cast<size_t>(nr) * (1 + nch) * w + static_cast<size_t>(F.n_s) * (3 + nch);
          auto U = [&](uint32_t row, uint32_t c) { return lds[F.base0 + c * nr - (c * (c - 1)) / 2 + (row - c)]; };
          for (int c = static_cast<int>(w) - 1; c >= 0; --c) {
            double dot = 0.0;
            for (uint32_t a = 0; a < rr; ++a) dot += U(w + a, c) * lds[xr[a] / 8u];
            for (uint32_t k = c + 1; k < w; ++k) dot += U(k, c) * lds[off_x + F.col0 + k];
            lds[off_x + F.col0 + c] = (U(nr - 1, c) - dot) * lds[off_invd + F.col0 + c];
          }
        }
      for (uint32_t i = 0; i < t.n_col; ++i) {
        const uint32_t pj = L.col_perm[t.col_off + i];
        h->xg[pj] = lds[off_x + i];
        h->p[L.perm[pj]] = lds[off_x + i];
      }
    }
  if (p_out) std::copy(h->p.begin(), h->p.end(), p_out);
}

// The dense plan (LdltPlan::dense; ldlt_dense_kernels.h): right-looking LDLT of the regularized matrix as a dense
// one in the natural order, the kernel's arithmetic (reciprocal of the pivot, fused multiply-add updates).
static void hc_factor_dense(hc_handle* h, double delta, double gamma) {
  const LdltPlan& L = h->l;
  const KktPlan& K = h->k;
  const int dim = L.n;
  std::vector<double>& A = h->dense_A;
  A.assign(static_cast<size_t>(dim) * dim, 0.0);
  for (int c = 0; c < dim; ++c) {
    for (int p = K.lhs.colptr[c]; p < K.lhs.colptr[c + 1]; ++p) A[static_cast<size_t>(c) * dim + K.lhs.rowidx[p]] += h->lhs[p];
    A[static_cast<size_t>(c) * dim + c] += c < L.n_dec ? delta : -gamma;
  }
  for (int k = 0; k < dim; ++k) {
    double* colk = A.data() + static_cast<size_t>(k) * dim;
    const double inv = 1.0 / colk[k];
    std::vector<double> u(colk + k + 1, colk + dim);
    for (int i = k + 1; i < dim; ++i) colk[i] = u[i - k - 1] * inv;
    for (int j = k + 1; j < dim; ++j) {
      double* colj = A.data() + static_cast<size_t>(j) * dim;
      for (int i = j; i < dim; ++i) colj[i] = std::fma(-colk[i], u[j - k - 1], colj[i]);
    }
  }
  for (int k = 0; k < dim; ++k) {
    const double u = A[static_cast<size_t>(k) * dim + k];
    h->D[k] = u;
    const double eps = 2.220446049250313e-16;
    if (u > eps) ++h->stats[0];
    else if (u < -eps) ++h->stats[1];
    else ++h->stats[2];
    if (u == 0.0 || !std::isfinite(u)) ++h->stats[3];
    else h->min_abs = std::min(h->min_abs, std::fabs(u));
    for (int i = k + 1; i < dim; ++i) h->Lx[L.Lp[k] + (i - k - 1)] = A[static_cast<size_t>(k) * dim + i];
  }
}
// ldlt_dense_pivoted_factor_kernel on the host: Eigen::LDLT's unblocked kernel (diagonal pivoting on the not yet updated
// diagonal, left-looking, sums in column order, no fused multiply-adds)
static void hc_factor_dense_pivoted(hc_handle* h, double delta, double gamma) {
  const LdltPlan& L = h->l;
  const KktPlan& K = h->k;
  const int dim = L.n;
  std::vector<double>& A = h->dense_A;
  A.assign(static_cast<size_t>(dim) * dim, 0.0);
  h->dense_trans.assign(dim, 0);
  auto at = [&](int r, int c) -> double& { return A[static_cast<size_t>(c) * dim + r]; };
  for (int c = 0; c < dim; ++c) {
    for (int p = K.lhs.colptr[c]; p < K.lhs.colptr[c + 1]; ++p) at(K.lhs.rowidx[p], c) += h->lhs[p];
    at(c, c) += c < L.n_dec ? delta : -gamma;
  }
  bool ret = true, found_zero = false;
  std::vector<double> temp(dim);
  for (int k = 0; k < dim; ++k) {
    int big = k;
    double bigv = std::fabs(at(k, k));
    for (int j = k + 1; j < dim; ++j)
      if (std::fabs(at(j, j)) > bigv) {
        bigv = std::fabs(at(j, j));
        big = j;
      }
    h->dense_trans[k] = big;
    if (big != k) {
      for (int c = 0; c < k; ++c) std::swap(at(k, c), at(big, c));
      for (int r = big + 1; r < dim; ++r) std::swap(at(r, k), at(r, big));
      for (int i = k + 1; i < big; ++i) std::swap(at(i, k), at(big, i));
      std::swap(at(k, k), at(big, big));
    }
    const int rs = dim - k - 1;
    if (k > 0) {
      for (int c = 0; c < k; ++c) temp[c] = at(c, c) * at(k, c);
      for (int r = -1; r < rs; ++r) {
        const int row = r < 0 ? k : k + 1 + r;
        volatile double sum = 0.0;  // (no contraction of the products into the sum)
        for (int c = 0; c < k; ++c) {
          const volatile double prod = at(row, c) * temp[c];
          sum = sum + prod;
        }
        at(row, k) -= sum;
      }
    }
    const double akk = at(k, k);
    const bool valid = std::fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < dim; ++j) {
        h->dense_trans[j] = j;
        for (int r = j + 1; r < dim; ++r) ret = ret && at(r, j) == 0.0;
      }
      break;
    }
    if (rs > 0 && valid) {
      for (int r = 0; r < rs; ++r) at(k + 1 + r, k) /= akk;
    } else if (rs > 0) {
      for (int r = 0; r < rs; ++r) ret = ret && at(k + 1 + r, k) == 0.0;
    }
    if (found_zero && valid) ret = false;
    else if (!valid) found_zero = true;
  }
  for (int k = 0; k < dim; ++k) {
    const double u = at(k, k);
    h->D[k] = u;
    const double eps = 2.220446049250313e-16;
    if (u > eps) ++h->stats[0];
    else if (u < -eps) ++h->stats[1];
    else ++h->stats[2];
    if (u != 0.0 && std::isfinite(u)) h->min_abs = std::min(h->min_abs, std::fabs(u));
    if (!std::isfinite(u)) ret = false;
  }
  if (!ret) ++h->stats[3];
}
static void hc_solve_dense(hc_handle* h, double* p_out) {
  const int dim = h->l.n;
  const std::vector<double>& A = h->dense_A;
  std::vector<double> x(h->rhs.begin(), h->rhs.begin() + dim);
  if (h->l.dense_pivoted) {
    for (int k = 0; k < dim; ++k) std::swap(x[k], x[h->dense_trans[k]]);
    for (int k = 0; k < dim; ++k)
      for (int i = k + 1; i < dim; ++i) x[i] = std::fma(-A[static_cast<size_t>(k) * dim + i], x[k], x[i]);
    for (int i = 0; i < dim; ++i) {
      const double d = A[static_cast<size_t>(i) * dim + i];
      x[i] = std::fabs(d) > 2.2250738585072014e-308 ? x[i] / d : 0.0;
    }
    for (int k = dim - 1; k >= 0; --k)
      for (int i = 0; i < k; ++i) x[i] = std::fma(-A[static_cast<size_t>(i) * dim + k], x[k], x[i]);
    for (int k = dim - 1; k >= 0; --k) std::swap(x[k], x[h->dense_trans[k]]);
    h->p = x;
    if (p_out) std::copy(h->p.begin(), h->p.end(), p_out);
    return;
  }
  for (int k = 0; k < dim; ++k)
    for (int i = k + 1; i < dim; ++i) x[i] = std::fma(-A[static_cast<size_t>(k) * dim + i], x[k], x[i]);
  for (int i = 0; i < dim; ++i) x[i] = x[i] / A[static_cast<size_t>(i) * dim + i];
  for (int k = dim - 1; k >= 0; --k)
    for (int i = 0; i < k; ++i) x[i] = std::fma(-A[static_cast<size_t>(i) * dim + k], x[k], x[i]);
  h->p = x;
  if (p_out) std::copy(h->p.begin(), h->p.end(), p_out);
}

void hc_factor(hc_handle* h, double delta, double gamma, double* D_out, double* stats_out) {
  const LdltPlan& L = h->l;
  std::memset(h->stats, 0, sizeof(h->stats));
  h->min_abs = INFINITY;
  if (L.dense || L.mf) {
    if (L.dense && L.dense_pivoted) hc_factor_dense_pivoted(h, delta, gamma);
    else if (L.dense) hc_factor_dense(h, delta, gamma);
    else hc_factor_mf(h, delta, gamma);
    if (D_out) std::copy(h->D.begin(), h->D.end(), D_out);
    if (stats_out) {
      for (int i = 0; i < 4; ++i) stats_out[i] = h->stats[i];
      stats_out[4] = h->min_abs;
    }
    return;
  }
  for (int r = 0; r < L.n_rounds; ++r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      std::vector<double> U(t.n_ent), invd(t.n_col);
      const uint32_t* lvl = L.lvl_ptr.data() + t.lvl_off;
      const uint32_t* pptr = L.ent_pair_ptr.data() + t.pair_ptr_off;
      const uint32_t* cptr = L.ent_contrib_ptr.data() + t.contrib_ptr_off;
      const LdltPair* pairs = L.pairs.data() + t.pair_off;
      const uint32_t* cidx = L.contrib_idx.data() + t.contrib_off;
      for (uint32_t l = 0; l < t.n_lvl; ++l) {
        std::vector<double> acc_v(lvl[l + 1] - lvl[l]);
        for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) {
          uint32_t e = t.ent_off + i;
          int32_t src = L.ent_src[e];
          uint8_t fl = L.ent_flags[e];
          double acc = src >= 0 ? ((fl & 4) ? h->rhs[src] : h->lhs[src]) : 0.0;  // bit 2: rhs row
          if (fl & 1) acc += (fl & 2) ? -gamma : delta;
          for (uint32_t c = cptr[i]; c < cptr[i + 1]; ++c) acc -= h->contrib[cidx[c]];
          for (uint32_t q = pptr[i]; q < pptr[i + 1]; ++q)
            acc -= (U[pairs[q].a] * invd[pairs[q].k]) * U[pairs[q].b];
          acc_v[i - lvl[l]] = acc;
        }
        for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) {
          uint32_t e = t.ent_off + i;
          U[i] = acc_v[i - lvl[l]];
          if ((L.ent_flags[e] & 9) == 1) invd[L.ent_col[e]] = 1.0 / U[i];
        }
        // supernodes of two or more columns in this level: the dense trapezoid (LdltSn),
        // eliminated column by column
        std::vector<uint32_t> descs;
        const uint32_t* snl = L.sn_lvl_ptr.data() + t.lvl_off;
        for (uint32_t q = snl[l]; q < snl[l + 1]; ++q) descs.push_back(q);
        for (uint32_t d : descs) {
          const LdltSn& sn = L.sn_desc[t.sn_off + d];
          auto off = [&](uint32_t c) { return sn.base0 + c * sn.nr - (c * (c - 1)) / 2; };
          for (uint32_t c = 0; c < sn.w; ++c) {
            const double inv = 1.0 / U[off(c)];
            invd[sn.col0 + c] = inv;
            for (uint32_t r = c + 1; r < sn.nr; ++r) {
              const double lrc = U[off(c) + (r - c)] * inv;
              for (uint32_t j = c + 1; j < sn.w && j <= r; ++j) U[off(j) + (r - j)] -= lrc * U[off(c) + (j - c)];
            }
          }
        }
      }
      for (uint32_t x = 0; x < t.n_ext; ++x) {
        double acc = 0.0;
        for (uint32_t q = pptr[t.n_ent + x]; q < pptr[t.n_ent + x + 1]; ++q)
          acc += (U[pairs[q].a] * invd[pairs[q].k]) * U[pairs[q].b];
        h->contrib[L.ext_dst[t.ext_off + x]] = acc;
      }
      for (uint32_t i = 0; i < t.n_ent; ++i) {
        uint32_t e = t.ent_off + i;
        double u = U[i];
        if (L.ent_flags[e] & 1) {
          h->D[L.ent_out[e]] = u;
          const double eps = 2.220446049250313e-16;
          if (u > eps) ++h->stats[0];
          else if (u < -eps) ++h->stats[1];
          else ++h->stats[2];
          if (u == 0.0 || !std::isfinite(u)) ++h->stats[3];
          else h->min_abs = std::min(h->min_abs, std::fabs(u));
        } else if (L.ent_flags[e] & 4) {
          h->z_factor[L.ent_out[e]] = u * invd[L.ent_col[e]];  // z = D^-1 L^-1 P b, by-product
        } else {
          h->Lx[L.ent_out[e]] = u * invd[L.ent_col[e]];
        }
      }
    }
  if (D_out) std::copy(h->D.begin(), h->D.end(), D_out);
  if (stats_out) {
    for (int i = 0; i < 4; ++i) stats_out[i] = h->stats[i];
    stats_out[4] = h->min_abs;
  }
}

static void hc_backward(hc_handle* h, double* p_out);

// backward substitution only, on the z the factorization left behind (the path the
// Newton step takes on the device)
void hc_solve_after_factor(hc_handle* h, double* p_out) {
  if (h->l.dense) {
    hc_solve_dense(h, p_out);
    return;
  }
  if (h->l.mf) {
    hc_backward_mf(h, p_out);
    return;
  }
  h->zv = h->z_factor;
  hc_backward(h, p_out);
}

// A new right-hand side through the fronts (ldlt_mf_kernels.h: ldlt_mf_solve_kernel): the factor in memory (Lx, D)
// back into every task's front layout, the forward substitution as the right-hand-side row of the factorization
// — per front the row's entries of the pivot columns, then its part of the update block —, then hc_backward_mf.
static void hc_forward_mf(hc_handle* h) {
  const LdltPlan& L = h->l;
  for (int r = 0; r < L.n_rounds; ++r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      const LdltMfTask& M = L.mf_tasks[ti];
      const uint32_t off_arena = t.n_ent, off_invd = off_arena + M.arena, off_x = off_invd + t.n_col;
      std::vector<double>& lds = h->mf_lds[ti];
      lds.assign(off_x + t.n_col + M.n_anc + 1, 0.0);
      auto at = [&](uint16_t byte_off) -> double& { return lds[byte_off / 8u]; };
      for (uint32_t i = 0; i < t.n_ent; ++i) {
        const uint32_t e = t.ent_off + i;
        const uint8_t fl = L.ent_flags[e];
        const uint32_t o = L.ent_out[e];
        if (fl & 1) {
          lds[i] = h->D[o];
          lds[off_invd + L.ent_col[e]] = 1.0 / h->D[o];
        } else if (fl & 4) {
          lds[i] = h->rhs[L.ent_src[e]];
        } else {
          lds[i] = h->Lx[o] * h->D[L.col_perm[t.col_off + L.ent_col[e]]];
        }
      }
      const uint32_t* cptr = L.mf_contrib_ptr.data() + M.contrib_ptr_off;
      const uint32_t* cidx = L.mf_contrib_idx.data() + M.contrib_off;
      const uint16_t* cent = L.mf_cent.data() + M.cent_off;
      for (uint32_t j = 0; j < M.n_cent; ++j) {
        if (!(L.ent_flags[t.ent_off + cent[j]] & 4)) continue;
        for (uint32_t c = cptr[j]; c < cptr[j + 1]; ++c) lds[cent[j]] -= h->mf_contrib[cidx[c]];
      }
      const uint32_t* lvl = L.mf_lvl_ptr.data() + t.lvl_off;
      const uint16_t* tab0 = L.mf_tab.data() + M.tab_off;
      const uint32_t* ext = L.mf_ext.data() + M.ext_off;
      for (uint32_t l = 0; l < t.n_lvl; ++l)
        for (uint32_t q = lvl[l]; q < lvl[l + 1]; ++q) {
          const LdltFront& F = L.mf_fronts[M.front_off + q];
          const uint32_t w = F.w, nr = F.nr, nch = F.nch, rr = nr - w - 1;
          const uint16_t* piv = tab0 + F.tab;
          const uint16_t* upd = piv + static_cast<size_t>(nr) * (1 + nch) * w;
          const uint16_t* prow = piv + static_cast<size_t>(nr - 1) * (1 + nch) * w;
          std::vector<double> li(w);
          for (uint32_t c = 0; c < w; ++c) {
            double v = at(prow[c]);
            for (uint32_t k = 1; k <= nch; ++k) v += at(prow[k * w + c]);
            for (uint32_t c2 = 0; c2 < c; ++c2) v = std::fma(-li[c2], at(piv[(static_cast<size_t>(c) * (1 + nch)) * w + c2]), v);
            at(prow[c]) = v;
            li[c] = v * lds[off_invd + F.col0 + c];
          }
          for (uint32_t b = 0; b < rr; ++b) {
            const uint16_t* row = upd + static_cast<size_t>(rr * (rr + 1) / 2 + b) * (3 + nch);
            double v = 0.0;
            for (uint32_t k = 0; k < nch; ++k) v += at(row[3 + k]);
            for (uint32_t c = 0; c < w; ++c) {
              const uint32_t coff = c * nr - (c * (c - 1)) / 2 - c;
              v = std::fma(-li[c], lds[row[2] / 8u + coff], v);
            }
            if (F.flags & 1) h->mf_contrib[ext[F.ext + row[0]]] = -v;
            else at(row[0]) = v;
          }
        }
    }
}

void hc_solve(hc_handle* h, double* p_out) {
  const LdltPlan& L = h->l;
  if (L.dense) {
    hc_solve_dense(h, p_out);
    return;
  }
  if (L.mf && std::getenv("SLPX_MF_SOLVE") == nullptr) {
    hc_forward_mf(h);
    hc_backward_mf(h, p_out);
    return;
  }
  for (int r = 0; r < L.n_rounds; ++r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      std::vector<double> y(t.n_col);
      const uint32_t* lvl = L.col_lvl_ptr.data() + t.lvl_off;
      const uint32_t* fptr = L.fwd_ptr.data() + t.colptr_off;
      const uint32_t* fcptr = L.fwd_contrib_ptr.data() + t.colptr_off;
      const LdltSolveItem* items = L.fwd_items.data() + t.fwd_item_off;
      const uint32_t* scidx = L.scontrib_idx.data() + t.scontrib_off;
      for (uint32_t l = 0; l < t.n_lvl; ++l) {
        std::vector<double> acc_v(lvl[l + 1] - lvl[l]);
        for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) {
          uint32_t pj = L.col_perm[t.col_off + i];
          double acc = h->rhs[L.perm[pj]];
          for (uint32_t c = fcptr[i]; c < fcptr[i + 1]; ++c) acc -= h->scontrib[scidx[c]];
          const uint32_t pos = L.col_sn[t.col_off + i] & 0xffu;  // the last `pos` items: the row's own chain
          for (uint32_t q = fptr[i]; q < fptr[i + 1] - pos; ++q) acc -= h->Lx[items[q].lpos] * y[items[q].ref];
          acc_v[i - lvl[l]] = acc;
        }
        for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) y[i] = acc_v[i - lvl[l]];
        const uint32_t* snl = L.sn_lvl_ptr.data() + t.lvl_off;
        for (uint32_t q = snl[l]; q < snl[l + 1]; ++q) {
          const uint32_t i0 = L.sn_desc[t.sn_off + q].col0, w = L.sn_desc[t.sn_off + q].w;
          for (uint32_t c = 1; c < w; ++c)
            for (uint32_t k = 0; k < c; ++k) y[i0 + c] -= h->Lx[items[fptr[i0 + c + 1] - c + k].lpos] * y[i0 + k];
        }
      }
      const uint32_t* sptr = L.sext_ptr.data() + t.sext_ptr_off;
      const LdltSolveItem* sitems = L.sext_items.data() + t.sext_item_off;
      for (uint32_t x = 0; x < t.n_sext; ++x) {
        double acc = 0.0;
        for (uint32_t q = sptr[x]; q < sptr[x + 1]; ++q) acc += h->Lx[sitems[q].lpos] * y[sitems[q].ref];
        h->scontrib[L.sext_dst[t.sext_off + x]] = acc;
      }
      for (uint32_t i = 0; i < t.n_col; ++i) {
        uint32_t pj = L.col_perm[t.col_off + i];
        h->zv[pj] = y[i] / h->D[pj];
      }
    }
  hc_backward(h, p_out);
}

static void hc_backward(hc_handle* h, double* p_out) {
  const LdltPlan& L = h->l;
  for (int r = L.n_rounds - 1; r >= 0; --r)
    for (uint32_t ti = L.round_ptr[r]; ti < L.round_ptr[r + 1]; ++ti) {
      const LdltTask& t = L.tasks[ti];
      std::vector<double> x(t.n_col);
      const uint32_t* lvl = L.col_lvl_ptr.data() + t.lvl_off;
      const uint32_t* bptr = L.bwd_ptr.data() + t.colptr_off;
      const LdltSolveItem* items = L.bwd_items.data() + t.bwd_item_off;
      for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
        std::vector<double> acc_v(lvl[l + 1] - lvl[l]);
        for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) {
          uint32_t pj = L.col_perm[t.col_off + i];
          double acc = h->zv[pj];
          const uint32_t cs = L.col_sn[t.col_off + i];  // the first w - pos - 1 items: the column's own chain
          for (uint32_t q = bptr[i] + ((cs >> 8) - (cs & 0xffu) - 1u); q < bptr[i + 1]; ++q) {
            uint32_t ref = items[q].ref;
            double xi = (ref & 0x80000000u) ? h->xg[ref & 0x7fffffffu] : x[ref];
            acc -= h->Lx[items[q].lpos] * xi;
          }
          acc_v[i - lvl[l]] = acc;
        }
        for (uint32_t i = lvl[l]; i < lvl[l + 1]; ++i) x[i] = acc_v[i - lvl[l]];
        const uint32_t* snl = L.sn_lvl_ptr.data() + t.lvl_off;
        for (uint32_t q = snl[l]; q < snl[l + 1]; ++q) {
          const uint32_t i0 = L.sn_desc[t.sn_off + q].col0, w = L.sn_desc[t.sn_off + q].w;
          for (int c = static_cast<int>(w) - 2; c >= 0; --c)
            for (uint32_t k = c + 1; k < w; ++k) x[i0 + c] -= h->Lx[items[bptr[i0 + c] + (k - c - 1)].lpos] * x[i0 + k];
        }
      }
      for (uint32_t i = 0; i < t.n_col; ++i) {
        uint32_t pj = L.col_perm[t.col_off + i];
        h->xg[pj] = x[i];
        h->p[L.perm[pj]] = x[i];
      }
    }
  if (p_out) std::copy(h->p.begin(), h->p.end(), p_out);
}

void hc_backsub(hc_handle* h, const double* s, const double* z, double mu, double* ps_out, double* pz_out) {
  const auto& K = h->k;
  const auto& st = h->s;
  for (int r = 0; r < K.m_i; ++r) {
    double aipx = 0.0;
    for (int q = K.ai_rowptr[r]; q < K.ai_rowptr[r + 1]; ++q) aipx += h->V[K.ai_src[q]] * h->p[K.ai_col[q]];
    double sinv = 1.0 / s[r];
    double p_s = (h->V[st.off_ci + r] - s[r]) + aipx;
    h->ps[r] = p_s;
    h->pz[r] = mu * sinv - z[r] - (sinv * z[r]) * p_s;
  }
  if (ps_out) std::copy(h->ps.begin(), h->ps.begin() + K.m_i, ps_out);
  if (pz_out) std::copy(h->pz.begin(), h->pz.begin() + K.m_i, pz_out);
}

}  // extern "C"

// Debug aid: distribution of update-block contributions per target entry
// out = {n_contrib slots, targets with >= 1, max per target, targets with > 32, sum over targets}
extern "C" void hc_contrib_stats(hc_handle* h, int64_t* out) {
  const LdltPlan& L = h->l;
  int64_t targets = 0, mx = 0, big = 0, total = 0;
  for (const LdltTask& t : L.tasks)
    for (uint32_t e = 0; e < t.n_ent; ++e) {
      const int64_t c = L.ent_contrib_ptr[t.contrib_ptr_off + e + 1] - L.ent_contrib_ptr[t.contrib_ptr_off + e];
      if (c > 0) ++targets;
      if (c > 32) ++big;
      mx = std::max(mx, c);
      total += c;
    }
  out[0] = L.n_contrib;
  out[1] = targets;
  out[2] = mx;
  out[3] = big;
  out[4] = total;
}

// Fundamental supernodes of L (column j + 1 joins column j's supernode when j's only etree
// child... i.e. parent[j] == j + 1 and pattern(j) = {j + 1} U pattern(j + 1)):
// out = {supernodes, widest (columns), longest column below the diagonal, columns in
//        supernodes >= 4 wide, columns in supernodes >= 16 wide}
extern "C" void hc_supernodes(hc_handle* h, int64_t* out) {
  const LdltPlan& L = h->l;
  const int n = L.n;
  std::vector<int> nchild(n, 0);
  for (int j = 0; j < n; ++j)
    if (L.parent[j] >= 0) ++nchild[L.parent[j]];
  int64_t count = 0, widest = 0, longest = 0, in4 = 0, in16 = 0;
  int j = 0;
  while (j < n) {
    int w = 1;
    while (j + w < n && L.parent[j + w - 1] == j + w && nchild[j + w] == 1 &&
           (L.Lp[j + w] - L.Lp[j + w - 1]) == (L.Lp[j + w + 1] - L.Lp[j + w]) + 1)
      ++w;
    ++count;
    widest = std::max<int64_t>(widest, w);
    if (w >= 4) in4 += w;
    if (w >= 16) in16 += w;
    for (int k = j; k < j + w; ++k) longest = std::max<int64_t>(longest, L.Lp[k + 1] - L.Lp[k]);
    j += w;
  }
  out[0] = count;
  out[1] = widest;
  out[2] = longest;
  out[3] = in4;
  out[4] = in16;
}

// Elimination tree and column counts of L (permuted space) for structure studies in tests.
extern "C" void hc_ldlt_tree(hc_handle* h, int32_t* parent, int32_t* colcount) {
  const LdltPlan& L = h->l;
  for (int j = 0; j < L.n; ++j) {
    parent[j] = L.parent[j];
    colcount[j] = L.Lp[j + 1] - L.Lp[j];
  }
}

// Supernodes of the plan: out[0] = count (singletons included), out[1] = widest, out[2] = levels on
// the critical path, out[3 + w] = supernodes of width w for w < cap - 3.
// the multifrontal plan (ldlt_symbolic.hpp: LdltFront): {built, fronts, fronts on the matrix cores, most children
// values per entry, widest front, most rows, explicit zeros of the relaxed supernodes (nnz(L) - exact),
// largest table bytes, largest arena doubles, update slots}
extern "C" void hc_mf_plan(hc_handle* h, int64_t* out) {
  const LdltPlan& L = h->l;
  for (int i = 0; i < 10; ++i) out[i] = 0;
  out[0] = L.mf ? 1 : 0;
  if (!L.mf) return;
  size_t fronts = 0, tab = 0, arena = 0;
  int widest = 0;
  for (size_t ti = 0; ti < L.tasks.size(); ++ti) {
    const LdltMfTask& M = L.mf_tasks[ti];
    fronts += M.n_front;
    tab = std::max<size_t>(tab, 2 * M.n_tab);
    arena = std::max<size_t>(arena, M.arena);
    for (uint32_t q = 0; q < M.n_front; ++q) widest = std::max<int>(widest, L.mf_fronts[M.front_off + q].w);
  }
  out[1] = static_cast<int64_t>(fronts);
  out[2] = L.mf_n_mfma;
  out[3] = L.mf_max_nch;
  out[4] = widest;
  out[5] = L.mf_max_front_rows;
  out[6] = L.nnzL;
  out[7] = static_cast<int64_t>(tab);
  out[8] = static_cast<int64_t>(arena);
  out[9] = L.mf_n_contrib;
}

// every front of the plan, 8 numbers each: task, round, level, w, nr, nch, n_s, flags (profiles/mf_front_stats.py)
extern "C" int32_t hc_mf_fronts(hc_handle* h, int32_t* out, int32_t cap_fronts) {
  const LdltPlan& L = h->l;
  if (!L.mf) return 0;
  int32_t n = 0;
  for (size_t ti = 0; ti < L.tasks.size(); ++ti) {
    const LdltTask& T = L.tasks[ti];
    const LdltMfTask& M = L.mf_tasks[ti];
    const uint32_t* lp = L.mf_lvl_ptr.data() + T.lvl_off;
    for (uint32_t l = 0; l < T.n_lvl; ++l)
      for (uint32_t q = lp[l]; q < lp[l + 1]; ++q) {
        const LdltFront& f = L.mf_fronts[M.front_off + q];
        if (n < cap_fronts) {
          int32_t* o = out + 8 * static_cast<size_t>(n);
          o[0] = static_cast<int32_t>(ti); o[1] = static_cast<int32_t>(T.round); o[2] = static_cast<int32_t>(l);
          o[3] = f.w; o[4] = f.nr; o[5] = f.nch; o[6] = f.n_s; o[7] = f.flags;
        }
        ++n;
      }
  }
  return n;
}

extern "C" void hc_supernode_plan(hc_handle* h, int64_t* out, int32_t cap) {
  const LdltPlan& L = h->l;
  for (int i = 0; i < cap; ++i) out[i] = 0;
  out[0] = L.n_supernodes;
  out[1] = L.widest_supernode;
  out[2] = L.critical_levels;
  for (size_t w = 0; w < L.sn_width_hist.size() && 3 + static_cast<int>(w) < cap; ++w) out[3 + w] = L.sn_width_hist[w];
}
