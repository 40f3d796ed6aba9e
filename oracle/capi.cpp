// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// C-ABI over the oracle so tests/ (ctypes) and bench.py's cpu_baseline leg can
// drive it.  Nothing here is part of the product.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "problems.hpp"

using namespace orc;

namespace {

std::vector<Expr*> g_exprs;

int reg(Expr* e) {
  g_exprs.push_back(e);
  return static_cast<int>(g_exprs.size()) - 1;
}

struct ProblemHandle {
  Problem problem;
  std::vector<int> dv_ids;
  // last evaluation (orc_problem_eval)
  Vec x, s, y, z;
  double mu = 0.0;
  double f = 0.0;
  Vec g, c_e, c_i, rhs, p, p_x, p_y, p_s, p_z, D;
  CSC A_e, A_i, H, lhs;
  std::vector<int> perm;
  double delta = 0.0, gamma = 0.0;
  int factorizations = 0, info = 0, nnzL = 0;
  std::unique_ptr<RegularizedLDLT> solver;
  SolveStats stats;
};

std::vector<std::unique_ptr<ProblemHandle>> g_problems;

const CSC* pick_csc(ProblemHandle& h, int which) {
  switch (which) {
    case 0: return &h.A_e;
    case 1: return &h.A_i;
    case 2: return &h.H;
    case 3: return &h.lhs;
    default: return nullptr;
  }
}
const Vec* pick_vec(ProblemHandle& h, int which) {
  switch (which) {
    case 0: return &h.g;
    case 1: return &h.c_e;
    case 2: return &h.c_i;
    case 3: return &h.rhs;
    case 4: return &h.p_x;
    case 5: return &h.p_y;
    case 6: return &h.p_s;
    case 7: return &h.p_z;
    case 8: return &h.D;
    case 9: return &h.p;
    default: return nullptr;
  }
}

}  // namespace

extern "C" {

void orc_reset() {
  g_problems.clear();
  g_exprs.clear();
  arena().reset();
}

int orc_var(double v) { return reg(decision_variable(v)); }
int orc_const(double v) { return reg(constant(v)); }

// op codes follow orc::Op
int orc_unary(int op, int a) {
  Expr* x = g_exprs[a];
  Expr* r = nullptr;
  switch (static_cast<Op>(op)) {
    case Op::NEG: r = neg(x); break;
    case Op::ABS: r = abs(x); break;
    case Op::SIGN: r = sign(x); break;
    case Op::SQRT: r = sqrt(x); break;
    case Op::CBRT: r = cbrt(x); break;
    case Op::EXP: r = exp(x); break;
    case Op::LOG: r = log(x); break;
    case Op::LOG10: r = log10(x); break;
    case Op::SIN: r = sin(x); break;
    case Op::COS: r = cos(x); break;
    case Op::TAN: r = tan(x); break;
    case Op::ASIN: r = asin(x); break;
    case Op::ACOS: r = acos(x); break;
    case Op::ATAN: r = atan(x); break;
    case Op::SINH: r = sinh(x); break;
    case Op::COSH: r = cosh(x); break;
    case Op::TANH: r = tanh(x); break;
    case Op::ERF: r = erf(x); break;
    default: return -1;
  }
  return reg(r);
}

int orc_binary(int op, int a, int b) {
  Expr* l = g_exprs[a];
  Expr* r = g_exprs[b];
  Expr* out = nullptr;
  switch (static_cast<Op>(op)) {
    case Op::ADD: out = add(l, r); break;
    case Op::SUB: out = sub(l, r); break;
    case Op::MUL: out = mul(l, r); break;
    case Op::DIV: out = div(l, r); break;
    case Op::POW: out = pow(l, r); break;
    case Op::ATAN2: out = atan2(l, r); break;
    case Op::HYPOT: out = hypot(l, r); break;
    case Op::MAX: out = max(l, r); break;
    case Op::MIN: out = min(l, r); break;
    default: return -1;
  }
  return reg(out);
}

double orc_value(int id) {
  Graph g = topological_sort(g_exprs[id]);
  update_values(g);
  return g_exprs[id]->val;
}
void orc_set_value(int id, double v) { g_exprs[id]->val = v; }
int orc_type(int id) { return static_cast<int>(g_exprs[id]->type); }
int orc_opcode(int id) { return static_cast<int>(g_exprs[id]->op); }
long orc_num_nodes() { return static_cast<long>(arena().nodes.size()); }

// Dense row-major (nr x n) Jacobian of rows wrt
void orc_jacobian(const int* rows, int nr, const int* wrt, int n, double* out) {
  std::vector<Expr*> r(nr), w(n);
  for (int i = 0; i < nr; ++i) r[i] = g_exprs[rows[i]];
  for (int i = 0; i < n; ++i) w[i] = g_exprs[wrt[i]];
  Jacobian J(r, w);
  const CSC& m = J.value();
  std::memset(out, 0, sizeof(double) * nr * n);
  for (int c = 0; c < m.cols; ++c)
    for (int p = m.colptr[c]; p < m.colptr[c + 1]; ++p) out[m.rowidx[p] * n + c] += m.val[p];
}

void orc_gradient(int f, const int* wrt, int n, double* out) { orc_jacobian(&f, 1, wrt, n, out); }

// Dense row-major (n x n) Hessian; lower != 0 keeps only the lower triangle
void orc_hessian(int f, const int* wrt, int n, int lower, double* out) {
  std::vector<Expr*> w(n);
  for (int i = 0; i < n; ++i) w[i] = g_exprs[wrt[i]];
  Hessian H(g_exprs[f], w, lower != 0);
  const CSC& m = H.value();
  std::memset(out, 0, sizeof(double) * n * n);
  for (int c = 0; c < m.cols; ++c)
    for (int p = m.colptr[c]; p < m.colptr[c + 1]; ++p) out[m.rowidx[p] * n + c] += m.val[p];
}

// Symbolic gradient (gradient_tree): returns expression ids (or -1 for null)
void orc_gradient_tree(int f, const int* wrt, int n, int* out_ids) {
  std::vector<Expr*> w(n);
  for (int i = 0; i < n; ++i) w[i] = g_exprs[wrt[i]];
  auto grad = gradient_tree(topological_sort(g_exprs[f]), w);
  for (int i = 0; i < n; ++i) out_ids[i] = grad[i] ? reg(grad[i]) : -1;
}

// ---------------------------------------------------------------------------
// Problem
// ---------------------------------------------------------------------------

int orc_problem_new() {
  g_problems.push_back(std::make_unique<ProblemHandle>());
  return static_cast<int>(g_problems.size()) - 1;
}

int orc_problem_decision_variable(int p) {
  Var v = g_problems[p]->problem.decision_variable();
  int id = reg(v.e);
  g_problems[p]->dv_ids.push_back(id);
  return id;
}
void orc_problem_minimize(int p, int id) { g_problems[p]->problem.minimize(Var(g_exprs[id])); }
void orc_problem_maximize(int p, int id) { g_problems[p]->problem.maximize(Var(g_exprs[id])); }
void orc_problem_subject_to_eq(int p, int id) {
  g_problems[p]->problem.subject_to_eq({Var(g_exprs[id])});
}
void orc_problem_subject_to_ineq(int p, int id) {
  g_problems[p]->problem.subject_to_ineq({Var(g_exprs[id])});
}
int orc_problem_cost_type(int p) { return g_problems[p]->problem.cost_function_type(); }
int orc_problem_eq_type(int p) { return g_problems[p]->problem.equality_constraint_type(); }
int orc_problem_ineq_type(int p) { return g_problems[p]->problem.inequality_constraint_type(); }

void orc_problem_dims(int p, int* n, int* m_e, int* m_i) {
  auto& pr = g_problems[p]->problem;
  *n = pr.num_decision_variables();
  *m_e = pr.num_equality_constraints();
  *m_i = pr.num_inequality_constraints();
}

void orc_problem_get_x(int p, double* out) {
  auto& dv = g_problems[p]->problem.decision_variables();
  for (size_t i = 0; i < dv.size(); ++i) out[i] = dv[i].e->val;
}
void orc_problem_set_x(int p, const double* in) {
  auto& dv = g_problems[p]->problem.decision_variables();
  for (size_t i = 0; i < dv.size(); ++i) dv[i].e->val = in[i];
}
void orc_problem_get_duals(int p, double* s, double* y, double* z) {
  auto& pr = g_problems[p]->problem;
  std::copy(pr.last_s.begin(), pr.last_s.end(), s);
  std::copy(pr.last_y.begin(), pr.last_y.end(), y);
  std::copy(pr.last_z.begin(), pr.last_z.end(), z);
}

// stats_out: [iterations, factorizations, solves, t_ad, t_build, t_decomp, t_solve,
//             t_linesearch, t_total]
int orc_problem_solve(int p, double tolerance, int max_iterations, double timeout,
                      const int* perm, int perm_len, double* stats_out) {
  Options opt;
  opt.tolerance = tolerance;
  opt.max_iterations = max_iterations;
  if (timeout > 0) opt.timeout = timeout;
  SolveStats st;
  std::vector<int> up;
  if (perm && perm_len > 0) up.assign(perm, perm + perm_len);
  ExitStatus e = g_problems[p]->problem.solve(opt, &st, up.empty() ? nullptr : &up);
  if (stats_out) {
    stats_out[0] = st.iterations;
    stats_out[1] = st.factorizations;
    stats_out[2] = st.solves;
    stats_out[3] = st.t_ad;
    stats_out[4] = st.t_build;
    stats_out[5] = st.t_decomp;
    stats_out[6] = st.t_solve;
    stats_out[7] = st.t_linesearch;
    stats_out[8] = st.t_total;
  }
  return static_cast<int>(e);
}

// Problem::solve with a record per iteration (taken where the reference calls the user's
// iteration callbacks, interior_point.hpp:411-424; inside feasibility restoration the iterate is
// the restoration model's, its first n / m_i entries the outer x / s):
// out[r] = {iteration, len(x), |x[0:n]|_2, |s[0:m_i]|_2, |y|_2, |z|_2}
int orc_problem_solve_trace(int p, double tolerance, int max_iterations, const int* perm, int perm_len,
                            int max_records, double* out, int* n_records) {
  Options opt;
  opt.tolerance = tolerance;
  opt.max_iterations = max_iterations;
  std::vector<int> up;
  if (perm && perm_len > 0) up.assign(perm, perm + perm_len);
  auto& prob = g_problems[p]->problem;
  prob.ensure_setup();
  const int n = prob.evaluators()->callbacks.num_decision_variables;
  const int m_i = prob.evaluators()->callbacks.num_inequality_constraints;
  int count = 0;
  auto norm = [](const Vec& v, size_t len) {
    double a = 0.0;
    for (size_t i = 0; i < std::min(len, v.size()); ++i) a += v[i] * v[i];
    return std::sqrt(a);
  };
  prob.add_callback([&](const IterationInfo& it) {
    if (count < max_records) {
      double* r = out + 6 * count;
      r[0] = it.iteration;
      r[1] = static_cast<double>(it.x.size());
      r[2] = norm(it.x, n);
      r[3] = norm(it.s, m_i);
      r[4] = norm(it.y, it.y.size());
      r[5] = norm(it.z, it.z.size());
    }
    ++count;
    return false;
  });
  SolveStats st;
  ExitStatus e = prob.solve(opt, &st, up.empty() ? nullptr : &up);
  prob.clear_callbacks();
  if (n_records) *n_records = std::min(count, max_records);
  return static_cast<int>(e);
}

// feasibility_restoration (feasibility_restoration.hpp:347-628) from a caller-given iterate: `steps`
// iterations of the restoration problem's interior-point loop, left through the callback exit
// (:729-752), multipliers re-estimated (lagrange_multiplier_estimate.hpp:56-133).  x, s, y, z in/out.
int orc_problem_restoration_steps(int p, double tolerance, int max_iterations, double* x, double* s,
                                  double* y, double* z, double mu, int steps) {
  auto& prob = g_problems[p]->problem;
  prob.ensure_setup();
  auto& cb = prob.evaluators()->callbacks;
  const int n = cb.num_decision_variables, m_e = cb.num_equality_constraints, m_i = cb.num_inequality_constraints;
  Options opt;
  opt.tolerance = tolerance;
  opt.max_iterations = max_iterations;
  Vec vx(x, x + n), vs(s, s + m_i), vy(y, y + m_e), vz(z, z + m_i);
  int iterations = 0;
  std::vector<IterationCallback> stop{[steps](const IterationInfo& it) { return it.iteration >= steps; }};
  const ExitStatus e = feasibility_restoration(cb, stop, opt, vx, vs, vy, vz, mu, iterations, nullptr);
  std::copy(vx.begin(), vx.end(), x);
  std::copy(vs.begin(), vs.end(), s);
  std::copy(vy.begin(), vy.end(), y);
  std::copy(vz.begin(), vz.end(), z);
  return static_cast<int>(e);
}

int orc_build_cart_pole(int N, double dt) {
  int p = orc_problem_new();
  auto& h = *g_problems[p];
  CartPole tmp;
  build_cart_pole(tmp, dt, N);
  h.problem = std::move(tmp.problem);
  return p;
}

int orc_build_flywheel(int N, double dt) {
  int p = orc_problem_new();
  auto& h = *g_problems[p];
  Flywheel tmp;
  build_flywheel(tmp, dt, N);
  h.problem = std::move(tmp.problem);
  return p;
}

// Scaling chosen at setup (problem.hpp:615-616)
void orc_problem_scaling(int p, double* d_f, double* d_ce, double* d_ci) {
  auto& h = *g_problems[p];
  h.problem.ensure_setup();
  auto* ev = h.problem.evaluators();
  *d_f = ev->scaling.f;
  std::copy(ev->scaling.c_e.begin(), ev->scaling.c_e.end(), d_ce);
  std::copy(ev->scaling.c_i.begin(), ev->scaling.c_i.end(), d_ci);
}

// One Newton step (interior_point.hpp:426-482 after the callbacks at :245-251 /
// :809-812) at a caller-chosen state.  do_solve=0 stops after lhs/rhs.
// perm (optional) = fill-reducing permutation to use instead of the oracle's own.
// timing_out: [t_ad, t_build, t_decomp, t_solve] seconds
int orc_problem_newton_step(int p, const double* x, const double* s, const double* y,
                            const double* z, double mu, int do_solve, const int* perm,
                            int perm_len, int reuse_solver, double* timing_out) {
  using clock = std::chrono::steady_clock;
  auto& h = *g_problems[p];
  h.problem.ensure_setup();
  auto* ev = h.problem.evaluators();
  auto& cb = ev->callbacks;
  int n = cb.num_decision_variables, m_e = cb.num_equality_constraints,
      m_i = cb.num_inequality_constraints;
  h.x.assign(x, x + n);
  h.s.assign(s, s + m_i);
  h.y.assign(y, y + m_e);
  h.z.assign(z, z + m_i);
  h.mu = mu;
  auto t0 = clock::now();
  h.f = cb.f(h.x);
  h.c_e = cb.c_e(h.x);
  h.c_i = cb.c_i(h.x);
  h.A_e = cb.A_e(h.x);
  h.A_i = cb.A_i(h.x);
  h.g = cb.g(h.x);
  h.H = cb.H(h.x, h.y, h.z);
  auto t1 = clock::now();
  h.lhs = build_kkt_lhs(h.H, h.A_e, h.A_i, h.s, h.z);
  h.rhs = build_kkt_rhs(h.g, h.A_e, h.A_i, h.c_e, h.c_i, h.s, h.y, h.z, mu);
  auto t2 = clock::now();
  auto t3 = t2, t4 = t2;
  h.info = 0;
  if (do_solve) {
    if (!h.solver || !reuse_solver) {
      int lhs_rows = n + m_e;
      int AiTAi = lower_triangle(multiply(transpose(h.A_i), h.A_i)).nnz();
      bool sparse = double(h.H.nnz() + AiTAi + h.A_e.nnz()) < 0.25 * double(lhs_rows) * lhs_rows;
      h.solver = std::make_unique<RegularizedLDLT>(sparse, n, m_e, 1e-10);
      if (perm && perm_len > 0) h.solver->set_permutation(std::vector<int>(perm, perm + perm_len));
    }
    if (reuse_solver == 2) h.solver->forget_regularization();
    t2 = clock::now();
    h.solver->compute(h.lhs);
    t3 = clock::now();
    h.info = h.solver->info();
    h.delta = h.solver->hessian_regularization();
    h.gamma = h.solver->constraint_jacobian_regularization();
    h.factorizations = h.solver->factorizations();
    h.D = h.solver->vecD();
    h.nnzL = h.solver->sparse_solver().nnzL();
    h.perm = h.solver->sparse_solver().perm;
    if (h.info == 0) {
      h.p = h.solver->solve(h.rhs);
      back_substitute(h.p, h.A_i, vsub(h.c_i, h.s), h.s, h.z, mu, n, m_e, h.p_x, h.p_y, h.p_s,
                      h.p_z);
    }
    t4 = clock::now();
  }
  if (timing_out) {
    auto d = [](clock::time_point a, clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    timing_out[0] = d(t0, t1);
    timing_out[1] = d(t1, t2);
    timing_out[2] = d(t2, t3);
    timing_out[3] = d(t3, t4);
  }
  return h.info;
}

double orc_eval_f(int p) { return g_problems[p]->f; }
void orc_eval_reg(int p, double* delta, double* gamma, int* factorizations, int* nnzL) {
  auto& h = *g_problems[p];
  *delta = h.delta;
  *gamma = h.gamma;
  *factorizations = h.factorizations;
  *nnzL = h.nnzL;
}
int orc_eval_csc_nnz(int p, int which) { return pick_csc(*g_problems[p], which)->nnz(); }
void orc_eval_csc(int p, int which, int* colptr, int* rowidx, double* val) {
  const CSC* m = pick_csc(*g_problems[p], which);
  std::copy(m->colptr.begin(), m->colptr.end(), colptr);
  std::copy(m->rowidx.begin(), m->rowidx.end(), rowidx);
  std::copy(m->val.begin(), m->val.end(), val);
}
int orc_eval_vec_len(int p, int which) {
  return static_cast<int>(pick_vec(*g_problems[p], which)->size());
}
void orc_eval_vec(int p, int which, double* out) {
  const Vec* v = pick_vec(*g_problems[p], which);
  std::copy(v->begin(), v->end(), out);
}
void orc_eval_perm(int p, int* out) {
  auto& h = *g_problems[p];
  std::copy(h.perm.begin(), h.perm.end(), out);
}

// Standalone regularized LDLᵀ on a caller-provided lower-triangular CSC
// (sparse_regularized_ldlt.hpp:64-161) — used by the LDLᵀ parity tests.
int orc_ldlt_solve(int n_total, int n, int m_e, const int* colptr, const int* rowidx,
                   const double* val, const double* rhs, const int* perm, int perm_len,
                   double gamma_min, double prev_delta_unused, double* x_out, double* D_out,
                   double* delta_gamma_out, int* factorizations_out) {
  (void)prev_delta_unused;
  CSC lhs(n_total, n_total);
  lhs.colptr.assign(colptr, colptr + n_total + 1);
  lhs.rowidx.assign(rowidx, rowidx + colptr[n_total]);
  lhs.val.assign(val, val + colptr[n_total]);
  RegularizedLDLT solver(true, n, m_e, gamma_min);
  if (perm && perm_len > 0) solver.set_permutation(std::vector<int>(perm, perm + perm_len));
  solver.compute(lhs);
  if (delta_gamma_out) {
    delta_gamma_out[0] = solver.hessian_regularization();
    delta_gamma_out[1] = solver.constraint_jacobian_regularization();
  }
  if (factorizations_out) *factorizations_out = solver.factorizations();
  Vec D = solver.vecD();
  if (D_out) std::copy(D.begin(), D.end(), D_out);
  if (solver.info() == Success && rhs && x_out) {
    Vec b(rhs, rhs + n_total);
    Vec x = solver.solve(b);
    std::copy(x.begin(), x.end(), x_out);
  }
  return solver.info();
}

// One unregularized/regularized numeric factorization with explicit (delta, gamma):
// returns info; D_out in the ORACLE's pivot order (perm order).
int orc_ldlt_factor_once(int n_total, int n, int m_e, const int* colptr, const int* rowidx,
                         const double* val, const int* perm, int perm_len, double delta,
                         double gamma, double* D_out, int* inertia_out) {
  CSC lhs(n_total, n_total);
  lhs.colptr.assign(colptr, colptr + n_total + 1);
  lhs.rowidx.assign(rowidx, rowidx + colptr[n_total]);
  lhs.val.assign(val, val + colptr[n_total]);
  RegularizedLDLT helper(true, n, m_e, 0.0);
  CSC a = add(lhs, helper.regularization(delta, gamma));
  SimplicialLDLT s;
  if (perm && perm_len > 0) s.set_permutation(std::vector<int>(perm, perm + perm_len));
  s.analyze_pattern(a);
  Info info = s.factorize(a);
  if (D_out) std::copy(s.D.begin(), s.D.end(), D_out);
  if (inertia_out) {
    Inertia in(s.D);
    inertia_out[0] = in.positive;
    inertia_out[1] = in.negative;
    inertia_out[2] = in.zero;
  }
  return info;
}

}  // extern "C"
