// Implementation of include/slpx.h.
#include "../../include/slpx.h"

#include <chrono>
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <filesystem>

#include "capi_internal.hpp"
#include "ipm.hpp"
#include "tape_jit.hpp"

namespace {
thread_local std::string g_error;

template <typename F>
int guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -100;
  } catch (...) {
    g_error = "unknown error";
    return -100;
  }
}
}  // namespace

extern "C" {

int slpx_abi_version(void) { return SLPX_ABI_VERSION; }
const char* slpx_last_error(void) { return g_error.c_str(); }
int slpx_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

int slpx_shard_range(int64_t n_items, int32_t rank, int32_t world, int64_t* lo, int64_t* hi) {
  if (world < 1 || rank < 0 || rank >= world || n_items < 0) return -1;
  const int64_t base = n_items / world, extra = n_items % world;
  *lo = rank * base + std::min<int64_t>(rank, extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
  return 0;
}

void slpx_graph_reset(void) { slpx::graph().clear(); }
int64_t slpx_graph_size(void) { return static_cast<int64_t>(slpx::graph().size()); }
int32_t slpx_expr_variable(double value) { return slpx::graph().variable(value); }
int32_t slpx_expr_constant(double value) { return slpx::graph().constant(value); }
int32_t slpx_expr_unary(int op, int32_t a) {
  return slpx::graph().unary(static_cast<slpx::Opcode>(op), a);
}
int32_t slpx_expr_binary(int op, int32_t a, int32_t b) {
  return slpx::graph().binary(static_cast<slpx::Opcode>(op), a, b);
}
int slpx_expr_type(int32_t id) { return slpx::graph().type[id]; }
double slpx_expr_value(int32_t id) { return slpx::graph().value(id); }
void slpx_expr_set_value(int32_t id, double value) { slpx::graph().val[id] = value; }
void slpx_expr_gradient_tree(int32_t f, const int32_t* wrt, int32_t n, int32_t* out) {
  auto& g = slpx::graph();
  std::vector<slpx::NodeId> w(wrt, wrt + n);
  auto grad = g.gradient_tree(g.topological_sort(f), w);
  for (int32_t i = 0; i < n; ++i) out[i] = grad[i];
}

slpx_problem* slpx_problem_create(void) { return new slpx_problem(); }
void slpx_problem_destroy(slpx_problem* p) { delete p; }
int32_t slpx_problem_decision_variable(slpx_problem* p) { return p->problem.decision_variable().expr; }
void slpx_problem_adopt_variable(slpx_problem* p, int32_t var) {
  p->problem.adopt_decision_variable(slp::Variable<double>::wrap(var));
}
void slpx_problem_minimize(slpx_problem* p, int32_t cost) { p->problem.minimize(slp::Variable<double>::wrap(cost)); }
void slpx_problem_maximize(slpx_problem* p, int32_t objective) {
  p->problem.maximize(slp::Variable<double>::wrap(objective));
}
void slpx_problem_subject_to_eq(slpx_problem* p, int32_t c) {
  p->problem.subject_to(slp::EqualityConstraints<double>{std::vector<slp::VariableF64>{slp::Variable<double>::wrap(c)}});
}
void slpx_problem_subject_to_ineq(slpx_problem* p, int32_t c) {
  p->problem.subject_to(slp::InequalityConstraints<double>{std::vector<slp::VariableF64>{slp::Variable<double>::wrap(c)}});
}
int slpx_problem_cost_type(const slpx_problem* p) { return static_cast<int>(p->problem.cost_function_type()); }
int slpx_problem_eq_type(const slpx_problem* p) { return static_cast<int>(p->problem.equality_constraint_type()); }
int slpx_problem_ineq_type(const slpx_problem* p) { return static_cast<int>(p->problem.inequality_constraint_type()); }
void slpx_problem_dims(const slpx_problem* p, int32_t* n, int32_t* m_e, int32_t* m_i) {
  *n = static_cast<int32_t>(p->problem.decision_variables().size());
  *m_e = static_cast<int32_t>(p->problem.equality_constraints().size());
  *m_i = static_cast<int32_t>(p->problem.inequality_constraints().size());
}
void slpx_problem_get_x(const slpx_problem* p, double* x) {
  auto& dv = p->problem.decision_variables();
  for (size_t i = 0; i < dv.size(); ++i) x[i] = slpx::graph().val[dv[i].expr];
}
void slpx_problem_set_x(slpx_problem* p, const double* x) {
  auto& dv = p->problem.decision_variables();
  for (size_t i = 0; i < dv.size(); ++i) slpx::graph().val[dv[i].expr] = x[i];
}

int slpx_problem_solve(slpx_problem* p, const slpx_options* o, slpx_report* report) {
  // (the struct as it was before `spy` was appended: an older caller's is that short)
  return slpx_problem_solve_sized(p, o, static_cast<uint32_t>(offsetof(slpx_options, spy)), report);
}

int slpx_problem_solve_sized(slpx_problem* p, const slpx_options* o, uint32_t options_bytes, slpx_report* report) {
  const bool spy = o != nullptr && options_bytes >= offsetof(slpx_options, spy) + sizeof(int32_t) && o->spy != 0;
  int status = -100;
  int rc = guard([&] {
    slp::Options opt;
    if (o) {
      opt.tolerance = o->tolerance;
      opt.max_iterations = o->max_iterations;
      if (o->timeout > 0) opt.timeout = o->timeout;
      opt.feasible_ipm = o->feasible_ipm != 0;
      opt.diagnostics = o->diagnostics != 0;
    }
    auto t0 = std::chrono::steady_clock::now();
    // problem.hpp:304-313: a problem without cost and constraints is SUCCESS before anything is
    // built — nothing to compile, no device needed
    const bool nothing_to_do = p->problem.cost_function_type() <= slp::ExpressionType::CONSTANT &&
                               p->problem.equality_constraint_type() <= slp::ExpressionType::CONSTANT &&
                               p->problem.inequality_constraint_type() <= slp::ExpressionType::CONSTANT;
    if (!nothing_to_do) p->problem.compile();
    p->t_compile = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    status = static_cast<int>(p->problem.solve(opt, spy));
    if (report) {
      const auto& r = p->problem.report();
      *report = slpx_report{r.iterations,    r.factorizations, r.solves,       r.value_sweeps,
                            r.delta,         r.gamma,          r.final_error,  r.t_setup,
                            r.t_kkt_build,   r.t_kkt_decomp,   r.t_kkt_solve,  r.t_line_search,
                            r.t_ad_refresh,  r.t_total,        p->t_compile,   r.restorations,
                            r.restoration_iterations, r.t_restoration_setup, r.t_restoration};
    }
  });
  return rc == 0 ? status : rc;
}

int slpx_problem_restoration_steps(slpx_problem* p, const slpx_options* o, double* x, double* sv, double* y,
                                   double* z, double mu, int32_t steps) {
  int status = -100;
  const int rc = guard([&] {
    slp::Options opt;
    if (o) {
      opt.tolerance = o->tolerance;
      opt.max_iterations = o->max_iterations;
      if (o->timeout > 0) opt.timeout = o->timeout;
    }
    const size_t n = p->problem.decision_variables().size(), me = p->problem.equality_constraints().size(),
                 mi = p->problem.inequality_constraints().size();
    std::vector<double> vx(x, x + n), vs(sv, sv + mi), vy(y, y + me), vz(z, z + mi);
    status = static_cast<int>(p->problem.restoration_steps(opt, vx, vs, vy, vz, mu, steps));
    std::copy(vx.begin(), vx.end(), x);
    std::copy(vs.begin(), vs.end(), sv);
    std::copy(vy.begin(), vy.end(), y);
    std::copy(vz.begin(), vz.end(), z);
  });
  return rc == 0 ? status : rc;
}

int slpx_problem_add_callback(slpx_problem* p, slpx_iteration_callback callback, void* user) {
  return guard([&] {
    if (!callback) throw std::runtime_error("slpx_problem_add_callback: null callback");
    p->problem.add_callback([p, callback, user](const slpx::IterationInfo& it) -> bool {
      const auto& st = it.structure ? *it.structure : p->problem.compile().structure();
      slpx_iteration_info info{};
      info.in_restoration = it.in_restoration ? 1 : 0;
      info.iteration = it.iteration;
      info.n = st.n;
      info.m_e = st.m_e;
      info.m_i = st.m_i;
      info.x = it.x.data();
      info.s = it.s.data();
      info.y = it.y.data();
      info.z = it.z.data();
      info.V = it.V.data();
      const int off[8] = {st.off_f, st.off_ce, st.off_ci, st.off_g, st.off_Ae, st.off_Ai, st.off_Hf, st.off_Hc};
      for (int k = 0; k < 8; ++k) info.off[k] = off[k];
      return callback(&info, user) != 0;
    });
  });
}
int slpx_problem_clear_callbacks(slpx_problem* p) {
  return guard([&] { p->problem.clear_callbacks(); });
}
slpx_system* slpx_problem_system(slpx_problem* p) {
  const int rc = guard([&] {
    if (!p->borrowed) p->borrowed = std::make_unique<slpx_system>();
    p->borrowed->owner = &p->problem;
    p->borrowed->ref = &p->problem.compile();
  });
  return rc == 0 ? p->borrowed.get() : nullptr;
}

void slpx_problem_get_duals(const slpx_problem* p, double* s, double* y, double* z) {
  if (s) std::copy(p->problem.slack().begin(), p->problem.slack().end(), s);
  if (y) std::copy(p->problem.equality_duals().begin(), p->problem.equality_duals().end(), y);
  if (z) std::copy(p->problem.inequality_duals().begin(), p->problem.inequality_duals().end(), z);
}

int slpx_problem_prebuild_kernels(slpx_problem* p, const char* dir) {
  int bodies = -1;
  const int rc = guard([&] {
    std::vector<slpx::NodeId> xs, ce, ci;
    for (auto& v : p->problem.decision_variables()) xs.push_back(v.expr);
    for (auto& v : p->problem.equality_constraints()) ce.push_back(v.expr);
    for (auto& v : p->problem.inequality_constraints()) ci.push_back(v.expr);
    const slpx::NodeId f =
        p->problem.cost_function_type() == slp::ExpressionType::NONE ? slpx::kNull : p->problem.cost().expr;
    const slpx::NlpStructure st = slpx::build_nlp_structure(slpx::graph(), xs, f, ce, ci, slpx::TapeCompileOptions{});
    slpx::TapeJitOptions opt;
    opt.n_unscaled_inputs = static_cast<uint32_t>(st.n);
    std::string where = dir ? dir : "";
    if (where.empty()) {
      Dl_info info{};
      if (dladdr(reinterpret_cast<const void*>(&slpx_problem_prebuild_kernels), &info) == 0 || !info.dli_fname)
        throw std::runtime_error("slpx_problem_prebuild_kernels: cannot locate libslpx.so");
      where = (std::filesystem::path(info.dli_fname).parent_path() / "jit_cache").string();
    }
    std::string log;
    const int a = slpx::prebuild_tape_templates(st.full, opt, where, log);
    const int b = slpx::prebuild_tape_templates(st.values, opt, where, log);
    // (and the full sweep's kernel as a chained step launches it: DeviceNlp picks that variant for one
    // problem whose step kernel leaves the sweep room on the chip)
    slpx::TapeJitOptions copt = opt;
    copt.chain_mode = 1;
    const int c = slpx::prebuild_tape_templates(st.full, copt, where, log);
    if (a < 0 || b < 0 || c < 0) throw std::runtime_error("slpx_problem_prebuild_kernels: " + log);
    bodies = a + b;
    // (feasibility restoration has no kernels of its own to generate: it runs on this system, csrc/restoration.hpp)
  });
  return rc == 0 ? bodies : rc;
}

slpx_system* slpx_system_create(slpx_problem* p, int32_t batch, int32_t device, const int32_t* perm,
                                int32_t perm_len) {
  auto* s = new slpx_system();
  int rc = guard([&] {
    if (batch < 1) throw std::runtime_error("slpx_system_create: batch must be at least 1");
    slpx::NewtonOptions opt;
    opt.batch = batch;
    opt.device = device;
    std::vector<slpx::NodeId> xs, ce, ci;
    for (auto& v : p->problem.decision_variables()) xs.push_back(v.expr);
    for (auto& v : p->problem.equality_constraints()) ce.push_back(v.expr);
    for (auto& v : p->problem.inequality_constraints()) ci.push_back(v.expr);
    std::vector<int32_t> up;
    if (perm && perm_len > 0) up.assign(perm, perm + perm_len);
    slpx::NodeId f = slpx::kNull;
    // Problem keeps m_f private; rebuild the cost handle through its type/expr accessors
    f = p->problem.cost_function_type() == slp::ExpressionType::NONE ? slpx::kNull : p->problem.cost().expr;
    s->sys = std::make_unique<slpx::NewtonSystem>(slpx::graph(), xs, f, ce, ci, opt,
                                                  up.empty() ? nullptr : &up);
    s->ref = s->sys.get();
  });
  if (rc != 0) {
    delete s;
    return nullptr;
  }
  return s;
}
void slpx_system_destroy(slpx_system* s) { delete s; }
int slpx_system_set_stream(slpx_system* s, void* hip_stream) {
  return guard([&] { s->get().device().set_stream(static_cast<hipStream_t>(hip_stream)); });
}
int slpx_system_sync(slpx_system* s) {
  return guard([&] { SLPX_HIP_CHECK(hipStreamSynchronize(s->get().device().stream())); });
}

int slpx_system_info(const slpx_system* sc, int64_t* out) {
  auto* s = const_cast<slpx_system*>(sc);
  return guard([&] {
    const auto& st = s->get().structure();
    const auto& k = s->get().kkt();
    const auto& l = s->get().ldlt();
    std::memset(out, 0, sizeof(int64_t) * SLPX_INFO_COUNT);
    out[SLPX_INFO_N] = st.n;
    out[SLPX_INFO_ME] = st.m_e;
    out[SLPX_INFO_MI] = st.m_i;
    out[SLPX_INFO_NV] = st.nV;
    out[SLPX_INFO_NNZ_G] = st.g_pat.nnz();
    out[SLPX_INFO_NNZ_AE] = st.Ae.nnz();
    out[SLPX_INFO_NNZ_AI] = st.Ai.nnz();
    out[SLPX_INFO_NNZ_HF] = st.Hf.nnz();
    out[SLPX_INFO_NNZ_HC] = st.Hc.nnz();
    out[SLPX_INFO_NNZ_LHS] = k.lhs.nnz();
    out[SLPX_INFO_NNZ_L] = l.nnzL;
    out[SLPX_INFO_LDLT_ROUNDS] = l.n_rounds;
    out[SLPX_INFO_LDLT_TASKS] = static_cast<int64_t>(l.tasks.size());
    out[SLPX_INFO_ETREE_HEIGHT] = l.etree_height;
    out[SLPX_INFO_LDLT_PAIRS] = l.flops / 2;
    out[SLPX_INFO_TAPE_TASKS] = static_cast<int64_t>(st.full.tasks.size());
    out[SLPX_INFO_TAPE_NODES] = static_cast<int64_t>(st.full.total_nodes);
    out[SLPX_INFO_TAPE_SLOTS] = static_cast<int64_t>(st.full.total_slots);
    out[SLPX_INFO_TAPE_EDGES] = static_cast<int64_t>(st.full.total_edges);
    out[SLPX_INFO_TAPE_LEVELS] = st.full.max_levels;
    out[SLPX_INFO_TAPE_SLOT_LEVELS] = st.full.max_slot_levels;
    out[SLPX_INFO_ASSEMBLE_BYTES] = k.assemble_bytes;
    out[SLPX_INFO_RHS_BYTES] = k.rhs_bytes;
    out[SLPX_INFO_FACTOR_BYTES] = l.factor_bytes;
    out[SLPX_INFO_SOLVE_BYTES] = l.solve_bytes;
    // AD refresh lower bound: read x,y,z, write every dynamic V entry
    int64_t dyn = 0;
    for (uint8_t st_ : st.V_is_static) dyn += st_ ? 0 : 1;
    out[SLPX_INFO_SWEEP_BYTES] = 8LL * st.n_inputs() + 8LL * dyn;
    out[SLPX_INFO_STRUCT_SINGULAR] = l.structurally_singular_unregularized ? 1 : 0;
    out[SLPX_INFO_OFF_G] = st.off_g;
    out[SLPX_INFO_OFF_AE] = st.off_Ae;
    out[SLPX_INFO_OFF_AI] = st.off_Ai;
    out[SLPX_INFO_OFF_HF] = st.off_Hf;
    out[SLPX_INFO_OFF_HC] = st.off_Hc;
    out[SLPX_INFO_GRAPH_NODES] = static_cast<int64_t>(st.graph_nodes_after);
    out[SLPX_INFO_NONLINEAR_ROWS] = st.nonlinear_rows;
    out[SLPX_INFO_TAPE_GLOBAL_TASKS] = static_cast<int64_t>(st.full.global_tasks.size());
    out[SLPX_INFO_TAPE_SHARED_TASKS] = st.full.shared_tasks;
    // bytes of the compiled program as the LDS-staged kernel reads it (16-bit records,
    // unique templates only) + the per-task leaf / output maps
    const auto& f = st.full;
    out[SLPX_INFO_TAPE_PROGRAM_BYTES] = static_cast<int64_t>(
        2 * (f.node_rec16.size() + f.edges16.size() + f.slot_edge_ptr16.size()) +
        4 * (f.lvl_ptr.size() + f.slvl_ptr.size() + f.leaf_src.size() + 3 * f.vout_src.size() +
             3 * f.jout_slot.size()) +
        sizeof(slpx::TapeTask) * f.tasks.size());
    out[SLPX_INFO_LDLT_LEVELS] = l.critical_levels;
    out[SLPX_INFO_LDLT_SUPERNODES] = l.n_supernodes;
    out[SLPX_INFO_LDLT_WIDEST] = l.widest_supernode;
    out[SLPX_INFO_LDLT_MULTIFRONTAL] = l.mf ? 1 : 0;
    out[SLPX_INFO_LDLT_FRONTS] = l.mf ? static_cast<int64_t>(l.mf_fronts.size()) - 16 : 0;  // (16 padding records)
    out[SLPX_INFO_LDLT_MFMA_FRONTS] = l.mf ? l.mf_n_mfma : 0;
    out[SLPX_INFO_LDLT_DENSE] = l.dense ? (l.dense_pivoted ? 2 : 1) : 0;
  });
}

int32_t slpx_system_pattern(const slpx_system* sc, int which, int32_t* colptr, int32_t* rowidx) {
  auto* s = const_cast<slpx_system*>(sc);
  const auto& st = s->get().structure();
  const slpx::CscPattern* pat = nullptr;
  switch (which) {
    case 0: pat = &st.g_pat; break;
    case 1: pat = &st.Ae; break;
    case 2: pat = &st.Ai; break;
    case 3: pat = &st.Hf; break;
    case 4: pat = &st.Hc; break;
    case 5: pat = &s->get().kkt().lhs; break;
    default: return -1;
  }
  if (colptr) std::copy(pat->colptr.begin(), pat->colptr.end(), colptr);
  if (rowidx) std::copy(pat->rowidx.begin(), pat->rowidx.end(), rowidx);
  return pat->nnz();
}
int slpx_system_perm(const slpx_system* sc, int32_t* perm) {
  auto* s = const_cast<slpx_system*>(sc);
  const auto& l = s->get().ldlt();
  std::copy(l.perm.begin(), l.perm.end(), perm);
  return 0;
}

int slpx_system_set_scaling(slpx_system* s, const double* scales) {
  return guard([&] {
    const int ns = s->get().structure().n_scales();
    s->get().device().set_scaling(std::vector<double>(scales, scales + ns));
  });
}
int slpx_system_set_state(slpx_system* s, const double* x, const double* sl, const double* y,
                          const double* z, const double* mu) {
  return guard([&] {
    auto& dev = s->get().device();
    const auto& st = s->get().structure();
    if (x) dev.upload_x(x);
    if (sl || y || z) {
      // upload_duals wants all three; fetch missing ones from the device
      const size_t B = dev.batch();
      std::vector<double> hs(B * std::max(1, st.m_i)), hy(B * std::max(1, st.m_e)), hz(B * std::max(1, st.m_i));
      if (!sl && st.m_i) dev.download(dev.d_s(), hs.data(), B * st.m_i);
      if (!y && st.m_e) dev.download(dev.d_y(), hy.data(), B * st.m_e);
      if (!z && st.m_i) dev.download(dev.d_z(), hz.data(), B * st.m_i);
      dev.upload_duals(sl ? sl : hs.data(), y ? y : hy.data(), z ? z : hz.data());
      SLPX_HIP_CHECK(hipStreamSynchronize(dev.stream()));
    }
    if (mu) dev.upload_mu(mu);
    SLPX_HIP_CHECK(hipStreamSynchronize(dev.stream()));
  });
}

int slpx_tape_sweep(slpx_system* s, int full) {
  return guard([&] {
    if (full) s->get().device().sweep_full();
    else s->get().device().sweep_values();
  });
}
int slpx_kkt_assemble(slpx_system* s) { return guard([&] { s->get().device().assemble(); }); }
int slpx_kkt_rhs(slpx_system* s) { return guard([&] { s->get().device().build_rhs(); }); }

int slpx_ldlt_factor(slpx_system* s, const double* delta, const double* gamma, double* stats) {
  return guard([&] {
    auto& dev = s->get().device();
    const int B = dev.batch();
    std::vector<double> d(delta, delta + B), g(gamma, gamma + B);
    std::vector<uint8_t> active(B, 1);
    dev.factor(d, g, active);
    std::vector<slpx::LdltStats> st;
    dev.read_stats(st);
    for (int b = 0; b < B; ++b) {
      stats[5 * b + 0] = st[b].n_pos;
      stats[5 * b + 1] = st[b].n_neg;
      stats[5 * b + 2] = st[b].n_zero;
      stats[5 * b + 3] = st[b].n_bad;
      double m;
      std::memcpy(&m, &st[b].min_abs_bits, sizeof(m));
      stats[5 * b + 4] = m;
    }
  });
}
int slpx_ldlt_compute(slpx_system* s, int32_t* info, double* reg, int32_t* factorizations) {
  return guard([&] {
    auto res = s->get().compute();
    const int B = s->get().batch();
    for (int b = 0; b < B; ++b) {
      if (info) info[b] = static_cast<int32_t>(res[b]);
      if (reg) {
        reg[2 * b] = s->get().hessian_regularization()[b];
        reg[2 * b + 1] = s->get().constraint_jacobian_regularization()[b];
      }
    }
    if (factorizations) *factorizations = s->get().last_factorizations();
  });
}
int slpx_ldlt_reset(slpx_system* s, double gamma_min) {
  return guard([&] {
    s->get().reset_regularization();
    s->get().set_gamma_min(gamma_min);
  });
}
int slpx_ldlt_solve(slpx_system* s) { return guard([&] { s->get().device().solve(); }); }
int slpx_step_backsub(slpx_system* s) { return guard([&] { s->get().device().backsub(); }); }
int slpx_newton_step(slpx_system* s, int refresh_ad, int32_t* info) {
  return guard([&] {
    auto res = s->get().newton_step(refresh_ad != 0);
    if (info)
      for (size_t b = 0; b < res.size(); ++b) info[b] = static_cast<int32_t>(res[b]);
  });
}

int slpx_newton_steps(slpx_system* s, int32_t count, int refresh_ad, int forget_regularization, int32_t* info) {
  return guard([&] {
    auto& sys = s->get();
    std::vector<int32_t> worst(sys.batch(), 0);
    for (int32_t k = 0; k < count; ++k) {
      if (forget_regularization) sys.reset_regularization();
      auto res = sys.newton_step(refresh_ad != 0);
      for (size_t b = 0; b < res.size(); ++b) worst[b] |= static_cast<int32_t>(res[b]);
    }
    if (info) std::copy(worst.begin(), worst.end(), info);
  });
}

int slpx_system_regularization(slpx_system* s, double* reg) {
  return guard([&] {
    const int B = s->get().batch();
    for (int b = 0; b < B; ++b) {
      reg[2 * b] = s->get().hessian_regularization()[b];
      reg[2 * b + 1] = s->get().constraint_jacobian_regularization()[b];
    }
  });
}

int64_t slpx_system_get(slpx_system* s, int which, double* out) {
  int64_t count = -1;
  int rc = guard([&] {
    auto& dev = s->get().device();
    const auto& st = s->get().structure();
    const auto& k = s->get().kkt();
    const auto& l = s->get().ldlt();
    const int64_t B = dev.batch();
    const double* src = nullptr;
    switch (which) {
      case 0: src = dev.d_V(); count = B * st.nV; break;
      case 1: src = dev.d_lhs(); count = B * k.lhs.nnz(); break;
      case 2: src = dev.d_rhs(); count = B * k.dim; break;
      case 3: src = dev.d_p(); count = B * k.dim; break;
      case 4: src = dev.d_ps(); count = B * st.m_i; break;
      case 5: src = dev.d_pz(); count = B * st.m_i; break;
      case 6: dev.materialize_factor(); src = dev.d_D(); count = B * l.n; break;
      case 7: dev.materialize_factor(); src = dev.d_Lx(); count = B * l.nnzL; break;
      case 8: src = dev.d_x(); count = st.n; if (B != 1) throw std::runtime_error("slpx_system_get: x of a batch is strided"); break;
      case 9: src = dev.d_s(); count = B * st.m_i; break;
      case 10: src = dev.d_y(); count = B * st.m_e; break;
      case 11: src = dev.d_z(); count = B * st.m_i; break;
      default: throw std::runtime_error("slpx_system_get: bad selector");
    }
    if (out && count > 0) dev.download(src, out, static_cast<size_t>(count));
  });
  return rc == 0 ? count : rc;
}
int slpx_system_set_rhs(slpx_system* s, const double* rhs) {
  return guard([&] {
    auto& dev = s->get().device();
    const size_t count = static_cast<size_t>(dev.batch()) * s->get().kkt().dim;
    SLPX_HIP_CHECK(hipMemcpyAsync(dev.d_rhs(), rhs, count * sizeof(double), hipMemcpyHostToDevice, dev.stream()));
    SLPX_HIP_CHECK(hipStreamSynchronize(dev.stream()));
  });
}

int slpx_ipm_direction(slpx_system* s, double tau, double* out3) {
  return guard([&] {
    auto& dev = s->get().device();
    dev.ipm_enable();
    dev.ipm_direction(tau);
    dev.wait();
    const slpx::IpmDirOut& d = dev.ipm_host().dir;
    out3[0] = d.alpha_max;
    out3[1] = d.alpha_z;
    out3[2] = d.D_phi;
  });
}
int slpx_ipm_trial(slpx_system* s, double alpha, int s_from_ci, double* out4) {
  return guard([&] {
    auto& dev = s->get().device();
    dev.ipm_enable();
    dev.ipm_trial_point(alpha);
    dev.sweep_values_trial();
    dev.ipm_trial_metrics(alpha, s_from_ci != 0);
    dev.wait();
    const slpx::IpmTrialOut& t = dev.ipm_host().trial;
    out4[0] = t.f;
    out4[1] = t.viol;
    out4[2] = t.logsum;
    out4[3] = t.finite;
  });
}
int slpx_ipm_commit(slpx_system* s, double alpha, double alpha_z, int s_from_ci) {
  return guard([&] {
    auto& dev = s->get().device();
    dev.ipm_enable();
    dev.ipm_commit(alpha, alpha_z, s_from_ci != 0);
    dev.wait();
  });
}
int slpx_ipm_errors(slpx_system* s, const double* error_scales, double* out24) {
  return guard([&] {
    auto& dev = s->get().device();
    dev.ipm_enable();
    const int ns = s->get().structure().n_scales();
    dev.ipm_set_error_scaling(std::vector<double>(error_scales, error_scales + ns));
    dev.ipm_errors(false);
    dev.wait();
    const slpx::IpmErrOut& e = dev.ipm_host().err;
    static_assert(sizeof(slpx::IpmErrOut) == 24 * sizeof(double), "out24 mirrors IpmErrOut");
    std::memcpy(out24, &e, sizeof(e));
  });
}

slpx_system* slpx_ldlt_create(int32_t n, int32_t m_e, const int32_t* colptr, const int32_t* rowidx, int32_t batch,
                              int32_t device) {
  slpx_system* out = nullptr;
  guard([&] {
    slpx::CscPattern lower;
    lower.rows = lower.cols = n + m_e;
    lower.colptr.assign(colptr, colptr + n + m_e + 1);
    lower.rowidx.assign(rowidx, rowidx + colptr[n + m_e]);
    if (batch < 1) throw std::runtime_error("slpx_ldlt_create: batch must be at least 1");
    slpx::NewtonOptions opt;
    opt.batch = batch;
    opt.device = device;
    auto sys = std::make_unique<slpx_system>();
    sys->sys = std::make_unique<slpx::NewtonSystem>(lower, n, m_e, opt);
    sys->ref = sys->sys.get();
    out = sys.release();
  });
  return out;
}

int slpx_ldlt_set_matrix(slpx_system* s, const double* values) {
  return guard([&] {
    auto& sys = s->get();
    const std::vector<int32_t>& map = sys.user_lhs_map();
    if (map.empty()) throw std::runtime_error("slpx_ldlt_set_matrix: not a system made by slpx_ldlt_create");
    auto& dev = sys.device();
    const size_t B = dev.batch(), nnz = sys.kkt().lhs.nnz(), user_nnz = map.size();
    std::vector<double> lhs(B * nnz, 0.0);
    for (size_t b = 0; b < B; ++b)
      for (size_t k = 0; k < user_nnz; ++k) lhs[b * nnz + map[k]] = values[b * user_nnz + k];
    SLPX_HIP_CHECK(hipMemcpyAsync(dev.d_lhs(), lhs.data(), lhs.size() * sizeof(double), hipMemcpyHostToDevice, dev.stream()));
    SLPX_HIP_CHECK(hipStreamSynchronize(dev.stream()));
  });
}

int slpx_debug_tape_clocks(slpx_system* s, uint64_t* out16) {
  return guard([&] {
    unsigned long long t[16];
    s->get().device().debug_tape_clocks(t);
    for (int i = 0; i < 16; ++i) out16[i] = t[i];
  });
}

int slpx_debug_tmpl_clocks(slpx_system* s, uint64_t* out, int32_t blocks) {
  int n = -1;
  const int rc = guard([&] {
    n = s->get().device().debug_tmpl_clocks(reinterpret_cast<unsigned long long*>(out), blocks);
  });
  return rc == 0 ? n : rc;
}

int slpx_debug_ldlt_clocks(slpx_system* s, uint32_t next_round, uint64_t* out24) {
  return guard([&] {
    unsigned long long t[24];
    s->get().device().debug_ldlt_clocks(next_round, t);
    for (int i = 0; i < 24; ++i) out24[i] = t[i];
  });
}

int slpx_debug_chain(slpx_system* s, int action) {
  int n = -1;
  const int rc = guard([&] {
    auto& dev = s->get().device();
    if (action == 1) dev.debug_break_next_chain();
    n = dev.chain_failures();
  });
  return rc == 0 ? n : rc;
}

int slpx_system_set_lhs(slpx_system* s, const double* lhs) {
  return guard([&] {
    auto& dev = s->get().device();
    const size_t count = static_cast<size_t>(dev.batch()) * s->get().kkt().lhs.nnz();
    SLPX_HIP_CHECK(hipMemcpyAsync(dev.d_lhs(), lhs, count * sizeof(double), hipMemcpyHostToDevice, dev.stream()));
    SLPX_HIP_CHECK(hipStreamSynchronize(dev.stream()));
  });
}

int slpx_system_time_step(slpx_system* s, int iters, int refresh_ad, float* ms) {
  return guard([&] {
    auto& sys = s->get();
    auto& dev = sys.device();
    hipStream_t st = dev.stream();
    hipEvent_t e0, e1;
    SLPX_HIP_CHECK(hipEventCreate(&e0));
    SLPX_HIP_CHECK(hipEventCreate(&e1));
    const int B = dev.batch();
    // Every phase is enqueued `iters` times back to back between ONE pair of events, so the
    // per-launch figure is the kernel's duration (plus the ~1 us between dependent launches),
    // not the ~20 us an event pair around a single launch adds; this is the number that must
    // agree with the rocprofv3 --stats average of the same kernel.  All phases are
    // idempotent on the resident state.
    auto timed = [&](auto&& launch) {
      launch();  // warm
      SLPX_HIP_CHECK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) launch();
      SLPX_HIP_CHECK(hipEventRecord(e1, st));
      SLPX_HIP_CHECK(hipEventSynchronize(e1));
      float t = 0;
      SLPX_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
      return t / static_cast<float>(iters);
    };
    // the regularization the policy loop settles on for this state, and how many attempts it took
    sys.reset_regularization();
    if (refresh_ad) dev.sweep_full();
    dev.assemble();
    dev.build_rhs();
    sys.compute();
    const double nfact = sys.last_factorizations();
    const std::vector<double> delta = sys.hessian_regularization(), gamma = sys.constraint_jacobian_regularization();
    const std::vector<uint8_t> active(B, 1);

    if (B >= 16) {
      // A batch: the kernels run for tens of microseconds to milliseconds, an event pair per
      // launch costs nothing in comparison — and repeating a launch on the same few hundred
      // MB would let the 256 MB Infinity Cache serve part of the re-reads and overstate the
      // HBM rate.  So: the phases in step order, each launched once per iteration.
      hipEvent_t ev[7];
      for (auto& e : ev) SLPX_HIP_CHECK(hipEventCreate(&e));
      double acc[6] = {0, 0, 0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
        SLPX_HIP_CHECK(hipEventRecord(ev[0], st));
        if (refresh_ad) dev.sweep_full();
        SLPX_HIP_CHECK(hipEventRecord(ev[1], st));
        dev.assemble();
        SLPX_HIP_CHECK(hipEventRecord(ev[2], st));
        dev.build_rhs();
        SLPX_HIP_CHECK(hipEventRecord(ev[3], st));
        dev.factor(delta, gamma, active);
        SLPX_HIP_CHECK(hipEventRecord(ev[4], st));
        dev.solve_after_factor();
        SLPX_HIP_CHECK(hipEventRecord(ev[5], st));
        dev.backsub();
        SLPX_HIP_CHECK(hipEventRecord(ev[6], st));
        SLPX_HIP_CHECK(hipEventSynchronize(ev[6]));
        for (int k = 0; k < 6; ++k) {
          float t = 0;
          SLPX_HIP_CHECK(hipEventElapsedTime(&t, ev[k], ev[k + 1]));
          acc[k] += t;
        }
      }
      for (int k = 0; k < 6; ++k) ms[k] = static_cast<float>(acc[k] / iters);
      ms[3] *= static_cast<float>(nfact);
      for (auto& e : ev) (void)hipEventDestroy(e);
    } else {
      ms[0] = refresh_ad ? timed([&] { dev.sweep_full(); }) : 0.0f;
      ms[1] = timed([&] { dev.assemble(); });
      ms[2] = timed([&] { dev.build_rhs(); });
      ms[3] = static_cast<float>(nfact) * timed([&] { dev.factor(delta, gamma, active); });  // all attempts
      ms[4] = timed([&] { dev.solve_after_factor(); });  // the factorization carried the rhs: backward only
      ms[5] = timed([&] { dev.backsub(); });
    }
    ms[6] = ms[0] + ms[1] + ms[2] + ms[3] + ms[4] + ms[5];
    ms[7] = static_cast<float>(nfact);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

int slpx_system_time_fused_step(slpx_system* s, int iters, float* ms) {
  return guard([&] {
    slpx::NewtonSystem& sys = s->get();
    slpx::DeviceNlp& dev = sys.device();
    const int B = sys.options().batch;
    hipStream_t st = dev.stream();
    hipEvent_t e0, e1;
    SLPX_HIP_CHECK(hipEventCreate(&e0));
    SLPX_HIP_CHECK(hipEventCreate(&e1));
    auto timed = [&](auto&& launch) {
      launch();
      SLPX_HIP_CHECK(hipStreamSynchronize(st));
      SLPX_HIP_CHECK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) launch();
      SLPX_HIP_CHECK(hipEventRecord(e1, st));
      SLPX_HIP_CHECK(hipEventSynchronize(e1));
      float t = 0;
      SLPX_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
      return t / static_cast<float>(iters);
    };
    // the regularization the policy loop settles on for this state
    sys.reset_regularization();
    dev.sweep_full();
    dev.assemble();
    dev.build_rhs();
    sys.compute();
    const std::vector<double> delta = sys.hessian_regularization(), gamma = sys.constraint_jacobian_regularization();
    const std::vector<uint8_t> active(B, 1);
    ms[0] = timed([&] { dev.sweep_full(/*with_reduce=*/false); });
    ms[1] = timed([&] {
      dev.build_kkt_for_step(/*with_reduce=*/true);
      dev.factor_solve_publish(delta, gamma, active);
    });
    ms[2] = ms[0] + ms[1];
    ms[3] = dev.step_is_one_launch() ? (dev.step_is_multifrontal() ? 2.0f : 1.0f) : 0.0f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

}  // extern "C"
