/*
 * slpx — C-ABI of the MI355X-native interior-point Newton step.
 *
 * The reference (SleipnirGroup/Sleipnir) has no FFI boundary on this path: it is
 * a C++ header-template library.  The seam this library sits behind is made of
 * three reference C++ interfaces; every entry point below names
 * the reference interface it replaces.  Plain pointers and sizes only; opaque
 * handles; caller-owned host arrays; library-owned device memory; no exceptions
 * cross the ABI (functions return 0 on success, <0 on usage/HIP errors — see
 * slpx_last_error() — and >0 where documented).  One handle per host thread /
 * HIP stream, like the reference's thread_local pool (src/util/pool.cpp:5-8).
 *
 * All vectors are fp64, all indices int32, matrices are CSC (column-major
 * compressed, sorted row indices) with int32 indices — the layout of
 * Eigen::SparseMatrix<double> the reference passes around.  Batched buffers are
 * batch-major: buf[b * stride + i].
 *
 * There is NO CPU fallback: compute entry points fail with an error when no HIP
 * device is present.
 */
#ifndef SLPX_H_
#define SLPX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slpx_problem slpx_problem;
typedef struct slpx_system slpx_system;

/* ---- library ------------------------------------------------------------ */
/* ABI history.  4: slpx_options.spy appended.  5: slpx_problem_solve() reads the options struct as it
 * was up to version 3 and therefore IGNORES `spy` — a caller that sets it must call
 * slpx_problem_solve_sized(p, &opt, sizeof opt, &report) (present since version 5: check
 * slpx_abi_version() >= 5 before binding it); the two benchmark-model constructors of versions <= 4
 * (slpx_problem_cart_pole / _flywheel) are gone: a model is the CALLER's program, built through the
 * slp:: surface or slpx_expr_* / slpx_problem_* (tests/support/models/ holds the benchmarks' as fixtures).
 * 6: SLPX_INFO_LDLT_DENSE appended to the info array (SLPX_INFO_COUNT grew by one). */
#define SLPX_ABI_VERSION 6
int slpx_abi_version(void);
const char* slpx_last_error(void);
int slpx_device_count(void);

/* Multi-GPU: a batch of independent problems is sharded over one process per GPU,
 * contiguous blocks, the first n_items % world ranks one item more — how multistart hands whole
 * solves to threads (optimization/multistart.hpp:52-62).  *lo, *hi = the half-open range of
 * `rank`.  No collective is needed inside a Newton step; a host joins the ranks only for
 * end-of-region reductions (RCCL).  Returns 0, -1 on a bad rank. */
int slpx_shard_range(int64_t n_items, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);

/* ---- expression graph ------------------------------------------------------
 * Replaces: slp::Variable and the detail:: operator set
 * (include/sleipnir/autodiff/variable.hpp:52-295, expression.hpp:155-2080), i.e.
 * what python/cpp/autodiff/bind_variable.cpp binds for the reference's own FFI.
 * Op codes are slpx_op below.  Pruning/constant-folding/typing rules are the
 * reference's.  Ids index this thread's arena. */
typedef enum slpx_op {
  SLPX_OP_CONST = 0, SLPX_OP_VAR, SLPX_OP_ADD, SLPX_OP_SUB, SLPX_OP_NEG, SLPX_OP_MUL,
  SLPX_OP_DIV, SLPX_OP_POW, SLPX_OP_ABS, SLPX_OP_SIGN, SLPX_OP_SQRT, SLPX_OP_CBRT,
  SLPX_OP_EXP, SLPX_OP_LOG, SLPX_OP_LOG10, SLPX_OP_SIN, SLPX_OP_COS, SLPX_OP_TAN,
  SLPX_OP_ASIN, SLPX_OP_ACOS, SLPX_OP_ATAN, SLPX_OP_ATAN2, SLPX_OP_SINH, SLPX_OP_COSH,
  SLPX_OP_TANH, SLPX_OP_ERF, SLPX_OP_HYPOT, SLPX_OP_MAX, SLPX_OP_MIN, SLPX_OP_ISNONNEG,
  SLPX_OP_ISPOS
} slpx_op;

void slpx_graph_reset(void); /* frees this thread's arena; invalidates its problems */
int64_t slpx_graph_size(void);
int32_t slpx_expr_variable(double value);               /* Variable()          variable.hpp:289 */
int32_t slpx_expr_constant(double value);               /* Variable(double)    variable.hpp:64  */
int32_t slpx_expr_unary(int op, int32_t a);             /* sin(x), -x, ...                      */
int32_t slpx_expr_binary(int op, int32_t a, int32_t b); /* x*y, pow(x,y), ...                   */
int slpx_expr_type(int32_t id);                         /* Variable::type()    variable.hpp:153 */
double slpx_expr_value(int32_t id);                     /* Variable::value()   variable.hpp:143 */
void slpx_expr_set_value(int32_t id, double value);     /* Variable::set_value variable.hpp:125 */
/* Symbolic gradient: Gradient::get() / detail::gradient_tree (variable_matrix.hpp:1757-1805).
 * out[i] = expression id of d f / d wrt[i], or -1 when wrt[i] is unreachable from f. */
void slpx_expr_gradient_tree(int32_t f, const int32_t* wrt, int32_t n, int32_t* out);

/* ---- problem ------------------------------------------------------------------
 * Replaces: slp::Problem<double> (include/sleipnir/optimization/problem.hpp):
 * decision_variable :78, minimize :151, maximize :162, subject_to :196/:218,
 * cost_function_type :236, solve :281.  Constraint ids are `lhs - rhs`
 * expressions (variable.hpp:716-778); inequality convention c(x) >= 0. */
slpx_problem* slpx_problem_create(void);
void slpx_problem_destroy(slpx_problem* p);
int32_t slpx_problem_decision_variable(slpx_problem* p);
/* Registers an existing free variable (slpx_expr_variable) as a decision variable. */
void slpx_problem_adopt_variable(slpx_problem* p, int32_t var);
void slpx_problem_minimize(slpx_problem* p, int32_t cost);
void slpx_problem_maximize(slpx_problem* p, int32_t objective);
void slpx_problem_subject_to_eq(slpx_problem* p, int32_t c);
void slpx_problem_subject_to_ineq(slpx_problem* p, int32_t c);
int slpx_problem_cost_type(const slpx_problem* p);
int slpx_problem_eq_type(const slpx_problem* p);
int slpx_problem_ineq_type(const slpx_problem* p);
void slpx_problem_dims(const slpx_problem* p, int32_t* n, int32_t* m_e, int32_t* m_i);
void slpx_problem_get_x(const slpx_problem* p, double* x);
void slpx_problem_set_x(slpx_problem* p, const double* x);

/* slp::Options (solver/options.hpp:13-38) */
typedef struct slpx_options {
  double tolerance;    /* 1e-8 */
  int32_t max_iterations; /* 5000 */
  double timeout;      /* seconds; <= 0 means infinity */
  int32_t feasible_ipm;
  int32_t diagnostics;
  int32_t spy; /* Problem::solve(options, spy) (problem.hpp:281): write H.spy, A_e.spy, A_i.spy (util/spy.hpp) into the
                  working directory, one record per iteration; since ABI version 4.  Read ONLY through
                  slpx_problem_solve_sized: slpx_problem_solve takes the struct as it was up to ABI version 3
                  (a caller built against an older header passes a shorter one) */
} slpx_options;

/* Per-solve counters and wall-clock phases (names of interior_point.hpp:155-174) */
typedef struct slpx_report {
  int32_t iterations, factorizations, solves, value_sweeps;
  double delta, gamma, final_error;
  double t_setup, t_kkt_build, t_kkt_decomp, t_kkt_solve, t_line_search, t_ad_refresh, t_total;
  double t_compile; /* graph -> tape/KKT plan/symbolic LDLT + upload */
  int32_t restorations; /* feasibility-restoration phases entered (feasibility_restoration.hpp:347) */
  int32_t restoration_iterations; /* of `iterations`, those spent inside them */
  double t_restoration_setup; /* compiling the restoration system (first phase only); in t_total */
  double t_restoration;       /* the restoration iterations; in t_total */
} slpx_report;

/* Problem::add_callback / clear_callbacks (problem.hpp:690-712) with IterationInfo
 * (solver/iteration_info.hpp:13-41).  Called at the start of every iteration with the iterate
 * and the AD outputs the solver holds at that point: V = [f | c_e | c_i | g | A_e | A_i | H_f |
 * H_c], matrices as value arrays over the static CSC patterns (off[] = where each block
 * starts; patterns and counts from slpx_problem_system + slpx_system_pattern / _info).  The
 * Lagrangian Hessian of iteration_info.hpp:34 is H_f + H_c.  Return non-zero to stop the solve
 * with CALLBACK_REQUESTED_STOP (exit_status.hpp:17).  The pointers are valid during the call. */
typedef struct slpx_iteration_info {
  int32_t iteration;
  int32_t n, m_e, m_i; /* sizes of the arrays below */
  const double* x; /* n */
  const double* s; /* m_i */
  const double* y; /* m_e */
  const double* z; /* m_i */
  const double* V;
  int64_t off[8]; /* f, c_e, c_i, g, A_e, A_i, H_f, H_c */
  /* 1 inside feasibility restoration (feasibility_restoration.hpp:347-628): the arrays are then
   * the RESTORATION model's — n + 2 m_e + 2 m_i variables [x | p_e | n_e | p_i | n_i] and
   * m_i + 2 m_e + 2 m_i inequality rows, the user's x and s first — and n, m_e, m_i, off[] and V
   * describe that model (the reference calls the user's callbacks there too, :852) */
  int32_t in_restoration;
} slpx_iteration_info;
typedef int (*slpx_iteration_callback)(const slpx_iteration_info* info, void* user);
int slpx_problem_add_callback(slpx_problem* p, slpx_iteration_callback callback, void* user);
int slpx_problem_clear_callbacks(slpx_problem* p);
/* The system solve() runs on (compiled now if it was not yet); owned by the problem, do NOT
 * pass it to slpx_system_destroy.  NULL + slpx_last_error() without a HIP device. */
slpx_system* slpx_problem_system(slpx_problem* p);

/* Problem::solve.  Returns slp::ExitStatus (solver/exit_status.hpp:13-43):
 * 0 success, 1 callback stop, -1..-10 as in the reference; -100 on library error. */
int slpx_problem_solve(slpx_problem* p, const slpx_options* opt, slpx_report* report);
/* The same with the caller's sizeof(slpx_options): members beyond `options_bytes` keep their defaults
 * (spy, ABI version 4, is read only when the caller's struct holds it). */
int slpx_problem_solve_sized(slpx_problem* p, const slpx_options* opt, uint32_t options_bytes, slpx_report* report);
void slpx_problem_get_duals(const slpx_problem* p, double* s, double* y, double* z);
/* feasibility_restoration (solver/util/feasibility_restoration.hpp:347-628) on its own: from the
 * iterate (x[n], s[m_i], y[m_e], z[m_i], mu) — all in/out but mu — build the restoration model,
 * run `steps` iterations of its interior-point loop, leave it the way the reference does when its
 * acceptance callback fires (:729-752), and replace y, z by the least-squares multiplier estimate
 * (lagrange_multiplier_estimate.hpp:56-133).  Scaling as in slpx_problem_solve (at the variables'
 * current values).  Returns the ExitStatus of that function (0: restored), -100 on library error. */
int slpx_problem_restoration_steps(slpx_problem* p, const slpx_options* opt, double* x, double* s, double* y,
                                   double* z, double mu, int32_t steps);

/* Build-time helper (no reference counterpart; needs NO device): compiles the model's structure
 * on the host, generates its tape kernels and cross-compiles them for gfx950 with hipRTC into
 * `dir` (NULL: <directory of libslpx.so>/jit_cache, where slpx_system_create looks before it
 * compiles anything).  Returns the number of generated bodies, < 0 on failure. */
int slpx_problem_prebuild_kernels(slpx_problem* p, const char* dir);

/* ---- compiled Newton system ------------------------------------------------------
 * One problem structure compiled for one GPU, `batch` independent value sets.
 * Replaces, per Newton step:
 *   AD evaluator seam  Gradient/Jacobian/Hessian::value()
 *                      (autodiff/gradient.hpp:53, jacobian.hpp:134, hessian.hpp:132)
 *                      and the Problem lambdas problem.hpp:618-660
 *   KKT build          solver/interior_point.hpp:426-448
 *   linear solver seam RegularizedLDLT::compute/solve/info/hessian_regularization/
 *                      constraint_jacobian_regularization (util/regularized_ldlt.hpp:56-128)
 *   back-substitution  solver/interior_point.hpp:470-481 */
slpx_system* slpx_system_create(slpx_problem* p, int32_t batch, int32_t device,
                                const int32_t* perm, int32_t perm_len);
void slpx_system_destroy(slpx_system* s);
int slpx_system_set_stream(slpx_system* s, void* hip_stream);
int slpx_system_sync(slpx_system* s);

enum {
  SLPX_INFO_N = 0, SLPX_INFO_ME, SLPX_INFO_MI, SLPX_INFO_NV, SLPX_INFO_NNZ_G, SLPX_INFO_NNZ_AE,
  SLPX_INFO_NNZ_AI, SLPX_INFO_NNZ_HF, SLPX_INFO_NNZ_HC, SLPX_INFO_NNZ_LHS, SLPX_INFO_NNZ_L,
  SLPX_INFO_LDLT_ROUNDS, SLPX_INFO_LDLT_TASKS, SLPX_INFO_ETREE_HEIGHT, SLPX_INFO_LDLT_PAIRS,
  SLPX_INFO_TAPE_TASKS, SLPX_INFO_TAPE_NODES, SLPX_INFO_TAPE_SLOTS, SLPX_INFO_TAPE_EDGES,
  SLPX_INFO_TAPE_LEVELS, SLPX_INFO_TAPE_SLOT_LEVELS, SLPX_INFO_ASSEMBLE_BYTES,
  SLPX_INFO_RHS_BYTES, SLPX_INFO_FACTOR_BYTES, SLPX_INFO_SOLVE_BYTES, SLPX_INFO_SWEEP_BYTES,
  SLPX_INFO_STRUCT_SINGULAR, SLPX_INFO_OFF_G, SLPX_INFO_OFF_AE, SLPX_INFO_OFF_AI,
  SLPX_INFO_OFF_HF, SLPX_INFO_OFF_HC, SLPX_INFO_GRAPH_NODES, SLPX_INFO_NONLINEAR_ROWS,
  SLPX_INFO_TAPE_GLOBAL_TASKS, SLPX_INFO_TAPE_SHARED_TASKS, SLPX_INFO_TAPE_PROGRAM_BYTES,
  /* supernodal LDLT: levels on the critical path (sum over rounds of the deepest task),
   * supernodes (singletons included), widest supernode in columns */
  SLPX_INFO_LDLT_LEVELS, SLPX_INFO_LDLT_SUPERNODES, SLPX_INFO_LDLT_WIDEST,
  /* since ABI version 5: the multifrontal plan — 1 if the single-problem step runs on fronts, their
   * number, and how many of them send their update block through the matrix cores */
  SLPX_INFO_LDLT_MULTIFRONTAL, SLPX_INFO_LDLT_FRONTS, SLPX_INFO_LDLT_MFMA_FRONTS,
  /* since ABI version 6: non-zero if the system is factored as a dense matrix (the reference's dense branch,
   * util/dense_regularized_ldlt.hpp).  2: chosen by the reference's own rule (interior_point.hpp:340-352: the lower
   * triangle fills a quarter of the system or more) and factored with Eigen::LDLT's diagonal pivoting; 1: the plain
   * dense kernel, where a column of L does not fit a task of the sparse plan (or SLPX_DENSE=1) */
  SLPX_INFO_LDLT_DENSE,
  SLPX_INFO_COUNT
};
int slpx_system_info(const slpx_system* s, int64_t* out /* SLPX_INFO_COUNT */);

/* Static sparsity patterns (CSC).  which: 0 g (1 x n), 1 A_e, 2 A_i, 3 H_f (lower),
 * 4 H_c (lower), 5 KKT lhs (lower, full diagonal).  Pass NULL to query only nnz. */
int32_t slpx_system_pattern(const slpx_system* s, int which, int32_t* colptr, int32_t* rowidx);
int slpx_system_perm(const slpx_system* s, int32_t* perm); /* fill-reducing permutation, perm[new]=old */

/* scales = [d_f, d_ce(m_e), d_ci(m_i)] (util/problem_scaling.hpp:100-107) */
int slpx_system_set_scaling(slpx_system* s, const double* scales);
/* x[batch][n], s[batch][m_i], y[batch][m_e], z[batch][m_i], mu[batch] (any may be NULL = keep) */
int slpx_system_set_state(slpx_system* s, const double* x, const double* sl, const double* y,
                          const double* z, const double* mu);

int slpx_tape_sweep(slpx_system* s, int full); /* full=1: values + g, A_e, A_i, H; 0: f, c_e, c_i */
int slpx_kkt_assemble(slpx_system* s);
int slpx_kkt_rhs(slpx_system* s);
/* One numeric factorization with explicit per-problem (delta, gamma).
 * stats[batch][5] = {n_pos, n_neg, n_zero, n_bad, min|D|} */
int slpx_ldlt_factor(slpx_system* s, const double* delta, const double* gamma, double* stats);
/* The whole inertia-correcting loop of sparse_regularized_ldlt.hpp:64-152.
 * info[batch] (0 Success, 1 NumericalIssue), reg[batch][2] = {delta, gamma} used. */
int slpx_ldlt_compute(slpx_system* s, int32_t* info, double* reg, int32_t* factorizations);
int slpx_ldlt_reset(slpx_system* s, double gamma_min);
int slpx_ldlt_solve(slpx_system* s); /* rhs -> p, reusable after one compute */
int slpx_step_backsub(slpx_system* s);
/* AD refresh (optional) + assemble + rhs + compute + solve + backsub */
int slpx_newton_step(slpx_system* s, int refresh_ad, int32_t* info);
/* `count` such steps one after the other on the resident state, each waited for like a single
 * one (the host reads the inertia before launching the next) — the loop a C++ harness writes
 * around slpx_newton_step, minus a foreign-function call per step.  forget_regularization != 0
 * clears the delta/gamma memory before every step (slpx_ldlt_reset keeps gamma_min), so each
 * step starts like the first iteration of a solve.  info[batch] = OR over the steps. */
int slpx_newton_steps(slpx_system* s, int32_t count, int refresh_ad, int forget_regularization, int32_t* info);
/* reg[batch][2] = {hessian_regularization(), constraint_jacobian_regularization()}
 * (util/regularized_ldlt.hpp:111,122) the last compute / Newton step settled on, per problem. */
int slpx_system_regularization(slpx_system* s, double* reg);

/* Device -> host copies.  which: 0 V, 1 lhs, 2 rhs, 3 p, 4 p_s, 5 p_z, 6 D (pivot order), 7 L values,
 * 8 x, 9 s, 10 y, 11 z (the resident iterate) */
int64_t slpx_system_get(slpx_system* s, int which, double* out);
int slpx_system_set_rhs(slpx_system* s, const double* rhs);
/* RegularizedLDLT::compute(lhs) with a caller-assembled matrix (regularized_ldlt.hpp:72):
 * values in the order of pattern 5 (lower CSC, forced diagonal), [batch][nnz]. */
int slpx_system_set_lhs(slpx_system* s, const double* lhs);

/* ---- The linear-solver seam on its own (SURVEY.md §8b seam 3) ----
 * RegularizedLDLT<double>(use_sparse = true, n, m_e, gamma_min) (util/regularized_ldlt.hpp:45-51):
 * a system with no expression graph, made from the lower-triangular CSC pattern of the matrix
 * compute() will be given — the `lhs` of interior_point.hpp:434-440, same pattern on every call
 * (regularized_ldlt.hpp:66-68); diagonal entries may be absent, they are added like
 * sparse_regularized_ldlt.hpp:67 does.  The first n rows/columns are regularized with +delta, the
 * other m_e with -gamma (:217-224).  The handle serves
 *   slpx_ldlt_reset(gamma_min)                       the constructor's gamma_min      (:45-51)
 *   slpx_ldlt_set_matrix + slpx_ldlt_compute         compute(lhs), info()             (:72, :56)
 *   slpx_system_set_rhs + slpx_ldlt_solve + slpx_system_get(3)     solve(rhs)         (:87)
 * (reg[] of slpx_ldlt_compute = hessian_regularization(), constraint_jacobian_regularization(),
 * :111,:122).  NULL + slpx_last_error() on failure, e.g. without a HIP device. */
slpx_system* slpx_ldlt_create(int32_t n, int32_t m_e, const int32_t* colptr, const int32_t* rowidx,
                              int32_t batch, int32_t device);
/* values[batch][nnz] in the order of the pattern given to slpx_ldlt_create */
int slpx_ldlt_set_matrix(slpx_system* s, const double* values);

/* ---- The interior-point iteration AROUND the Newton step, on the resident iterate ----
 * (one problem per system).  What interior_point() does between two Newton steps with
 * O(n) host loops over Eigen vectors (interior_point.hpp:488-563, :775-832) as kernels on
 * the state already in HBM; only the scalars below cross PCIe.  slpx_problem_solve() runs
 * on these; a maintainer binding at the linear-solver or AD seam does not need them.
 *
 * slpx_ipm_direction: for the step left by slpx_newton_step —
 *   out[3] = {alpha_max = ftb(s, p_s, tau), alpha_z = ftb(z, p_z, tau)
 *             (util/fraction_to_the_boundary_rule.hpp:19-43),
 *             D_phi = g.p_x - mu sum(p_s / s) (interior_point.hpp:508-509)}
 * slpx_ipm_trial: trial x = x + alpha p_x, forward sweep of f, c_e, c_i there —
 *   out[4] = {f, |c_e|_1 + |c_i - s_trial|_1, sum ln s_trial, all finite (1/0)}
 *   (the filter entry of util/filter.hpp:30-60 is {f - mu out[2], out[1]});
 *   s_trial = s + alpha p_s, or the trial c_i when s_from_ci (feasible-IPM option, :520-526)
 * slpx_ipm_commit: x += alpha p_x, s := s_trial, y += alpha_z p_y, z += alpha_z p_z, then
 *   the z reset of interior_point.hpp:797-801 (mu = the value last set with set_state)
 * slpx_ipm_errors: at the resident iterate with V from the last FULL sweep; error_scales =
 *   [d_f, d_ce, d_ci] used for un-scaling (util/kkt_error.hpp:216-251) —
 *   out[24] = {un-scaled: |g - A_e^T y - A_i^T z|_inf, |s.z|_inf, |c_e|_inf, |c_i - s|_inf,
 *              |y|_1, |z|_1;  scaled: the same dual residual, min s.z, max s.z, |c_e|_inf,
 *              |c_i - s|_inf, |y|_1, |z|_1;  f, |c_e|_1 + |c_i - s|_1, sum ln s;
 *              |A_e^T c_e|_2^2, |c_e|_2^2, |A_i^T c_i^-|_2^2, |c_i^-|_2^2
 *              (util/is_locally_infeasible.hpp:17-60);  |x|_inf, |s|_inf, all finite,
 *              all c_i > 0}  (util/kkt_error.hpp:92-146 combines the first thirteen) */
int slpx_ipm_direction(slpx_system* s, double tau, double* out3);
int slpx_ipm_trial(slpx_system* s, double alpha, int s_from_ci, double* out4);
int slpx_ipm_commit(slpx_system* s, double alpha, double alpha_z, int s_from_ci);
int slpx_ipm_errors(slpx_system* s, const double* error_scales, double* out24);

/* Per-kernel-group durations with HIP events on the system's stream: every phase of the
 * step is enqueued `iters` times back to back between one pair of events (the phases are
 * idempotent on the resident state), so the figure is the kernel's own duration — the one
 * rocprofv3 --stats reports — without the ~20 us an event pair around a single launch adds.
 * (Batches of 16 or more: one launch per phase per iteration, events between the phases.)
 * ms[8] = {sweep, assemble, rhs, factor (x attempts the policy loop needs), backward solve,
 * backsub, sum, factorizations per step} */
int slpx_system_time_step(slpx_system* s, int iters, int refresh_ad, float* ms);

/* The same for the launches a single problem's Newton step actually makes (device.hpp: KktFuse,
 * BacksubFuse, ldlt_mf_step_kernel): ms[4] = {AD sweep launch, the launch that evaluates the
 * KKT system, factorizes, solves and back-substitutes (first attempt of the policy loop's
 * settled regularization), their sum, 1 if that single launch exists for this system — 0: the
 * step is made of the kernels slpx_system_time_step times and ms[1] is their sum} */
int slpx_system_time_fused_step(slpx_system* s, int iters, float* ms);

/* Profiling aid: wall_clock64() ticks (100 MHz) recorded by workgroup 0 of the last tape
 * sweep at {entry, staged, leaves, forward done, values out, adjoints done, exit};
 * out16[0..7] = 64-thread kernel, out16[8..15] = 256-thread kernel. */
int slpx_debug_tape_clocks(slpx_system* s, uint64_t* out16);
/* Same for the first task of one LDLT round: out24[0..7] factor {entry, staged, values
 * gathered, levels done, update blocks done, exit}, [8..15] forward solve, [16..23]
 * backward solve.  Returns what was recorded so far, then arms `next_round`. */
int slpx_debug_ldlt_clocks(slpx_system* s, uint32_t next_round, uint64_t* out24);
/* With SLPX_TAPE_JIT_CLOCKS=1 in the environment when the system was made: ticks of the first
 * `blocks` (<= 2048) workgroups of the generated tape kernel's last launch, out[8*blocks]: {entry,
 * exit, body entered, leaves loaded, forward part done, -, -, -} per workgroup.
 * Returns the number of workgroups that run generated bodies (the rest interpret), or -1. */
int slpx_debug_tmpl_clocks(slpx_system* s, uint64_t* out, int32_t blocks);
/* Test hook for chained steps (consecutive slpx_newton_step calls of one problem: the AD sweep and the
 * step kernel side by side, ordered through words in memory with BOUNDED spins).  action 1: the next
 * chained sweep is told to wait for a step kernel that does not exist — it gives up after its bound,
 * the step reports the failure instead of computing on a torn V, and the library redoes that step
 * unchained.  Returns the number of chain failures recovered from so far (action 0: just that). */
int slpx_debug_chain(slpx_system* s, int action);

#ifdef __cplusplus
}
#endif
#endif /* SLPX_H_ */
