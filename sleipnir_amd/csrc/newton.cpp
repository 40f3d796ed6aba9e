#include "newton.hpp"

#include "restoration.hpp"

#include "setup_timing.hpp"

#include <algorithm>
#include <future>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace slpx {

// The sparse plan, or — where a column of L does not fit the LDS of a task (a Hessian dense in hundreds of
// variables: single shooting), or SLPX_DENSE=1 asks for it — the dense one: the reference's dense branch
// (util/dense_regularized_ldlt.hpp, chosen there by density: interior_point.hpp:340-352) instead of a refusal.
static LdltPlan plan_or_dense(const CscPattern& lhs, int n_dec, const LdltOptions& lopt, const std::vector<int32_t>* user_perm,
                              const std::vector<uint8_t>* diag_has_source, int batch) {
  constexpr int kDenseMaxOrder = 8192;  // 512 MB of factors per problem, two columns in LDS
  // (and the whole batch's factors, batch x n x n doubles, within reach of one device: 64 GB)
  auto dense_plan = [&] {
    const double bytes = 8.0 * static_cast<double>(std::max(1, batch)) * lhs.cols * lhs.cols;
    if (bytes > 64.0 * (1u << 30))
      throw std::runtime_error("slpx: the dense factorization of " + std::to_string(batch) + " systems of order " + std::to_string(lhs.cols) +
                               " needs " + std::to_string(static_cast<long long>(bytes / (1u << 30))) +
                               " GB of factors; SLPX_DENSE=0 keeps the sparse plan (or refuses the model), a smaller batch fits");
    return build_dense_ldlt_plan(lhs, n_dec);
  };
  const char* env = std::getenv("SLPX_DENSE");
  if (env != nullptr && env[0] == '1' && lhs.cols <= kDenseMaxOrder) return dense_plan();
  try {
    return build_ldlt_plan(lhs, n_dec, lopt, user_perm, diag_has_source);
  } catch (const std::runtime_error& e) {
    if (!ldlt_plan_error_is_too_big(e) || lhs.cols > kDenseMaxOrder || (env != nullptr && env[0] == '0')) throw;
    if (std::getenv("SLPX_LDLT_VERBOSE"))
      std::fprintf(stderr, "ldlt: %s — the system of order %d is factored as a dense matrix\n", e.what(), lhs.cols);
    return dense_plan();
  }
}

NewtonSystem::NewtonSystem(Graph& g, const std::vector<NodeId>& x, NodeId f,
                           const std::vector<NodeId>& c_e, const std::vector<NodeId>& c_i,
                           const NewtonOptions& opt, const std::vector<int32_t>* user_perm, bool defer_device)
    : m_opt(opt), m_graph(&g), m_x_nodes(x), m_ce_nodes(c_e), m_ci_nodes(c_i) {
  SetupLap lap;
  // the HIP runtime comes up (context, first allocation: 50-700 ms in a fresh process) while the
  // host compiles the model
  std::future<void> device_job;
  if (!defer_device)
    device_job = std::async(std::launch::async, [device = opt.device] {
      if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr);
      (void)hipGetLastError();
    });
  // the KKT plan and the symbolic factorization (25 ms at N=1000) alongside the tape compiler
  auto plan_linear_algebra = [&](const NlpStructure& st) {
    m_k = build_kkt_plan(st);
    // which diagonal entries of the unregularized lhs have any source at all
    std::vector<uint8_t> diag_has_source(m_k.dim, 0);
    for (int c = 0; c < m_k.dim; ++c)
      for (int p = m_k.lhs.colptr[c]; p < m_k.lhs.colptr[c + 1]; ++p)
        if (m_k.lhs.rowidx[p] == c)
          diag_has_source[c] = (m_k.dptr[p + 1] > m_k.dptr[p]) || (m_k.pptr[p + 1] > m_k.pptr[p]);
    // One problem: big tasks = few rounds and levels (latency).  A batch is throughput bound
    // by how many tasks fit a CU's LDS at once: half-size tasks (~30 KB instead of ~60 KB)
    // put five instead of two workgroups on a CU and have fewer levels each.
    LdltOptions lopt = opt.ldlt;
    if (opt.batch >= 16 && lopt.task_entries == LdltOptions{}.task_entries) lopt.task_entries = 1024;
    // one lane per problem (ldlt_il_kernels.h): a task's values x 64 problems must fit LDS
    // (measured at 512 x N=1000, ms per factorization: 192 -> 0.55 but some problems then need a
    // second attempt, 384 -> 0.67, 768 -> 0.79, 1024 -> 1.25)
    // (below ~200 problems the rounds, not the chip, bound a factorization: fewer, bigger tasks — 64 x N=500:
    // 384 entries 242 k steps/s, 512: 251 k, 768: 247 k)
    if (DeviceNlp::interleaved_for(opt.batch) && lopt.task_entries >= 1024) lopt.task_entries = opt.batch < 192 ? 512 : 384;
    // the interleaved kernels walk column levels; supernodal levels are the per-task kernels'
    if (DeviceNlp::interleaved_for(opt.batch)) lopt.supernodal = false;
    // one problem: the multifrontal step (ldlt_mf_kernels.h) — every supernode a dense front, so a chain
    // of two columns already saves a level (the pair-list kernels' chain pass only paid from four)
    if (opt.batch == 1 && lopt.supernodal) {
      const char* env = std::getenv("SLPX_LDLT_MF");
      if (env == nullptr || env[0] != '0') {
        lopt.multifrontal = true;
        lopt.min_supernode_width = 2;
        lopt.relax_zeros = 8;
        lopt.balance_supernode_cuts = true;
        lopt.chain_from_deepest_child = true;
        lopt.chain_from_deepest_min_round = 0;
      }
    }
    if (const char* env = std::getenv("SLPX_RELAX_ZEROS")) lopt.relax_zeros = std::atoi(env);
    if (const char* env = std::getenv("SLPX_SN_MIN_WIDTH")) lopt.min_supernode_width = std::atoi(env);
    if (const char* env = std::getenv("SLPX_MFMA_MIN_ENTRIES")) lopt.mfma_min_entries = static_cast<uint32_t>(std::atoi(env));
    // One problem (or a handful: the same plan, so that a small batch and single problems agree to
    // the bit), smaller than the BASELINE horizon: smaller tasks (less plan to stage per
    // task, shorter level passes, more of the chip in the leaf round) — task size with the square
    // root of the system's order.  Measured (fused kernel, us): cart-pole N=300 1024 entries 47.9
    // vs 2048 54.6; N=500 1536 52.3 vs 2048 53.8; N=1000 2048 56.2 vs 1792 59.4.
    if (opt.batch < 16 && lopt.task_entries == LdltOptions{}.task_entries && m_k.dim < 9000) {
      const double scaled = 2048.0 * std::sqrt(static_cast<double>(m_k.dim) / 9000.0);
      lopt.task_entries = std::clamp<uint32_t>(256u * static_cast<uint32_t>(std::lround(scaled / 256.0)), 1024u, 2048u);
    }
    if (const char* env = std::getenv("SLPX_TASK_ENTRIES")) lopt.task_entries = static_cast<uint32_t>(std::atoi(env));
    // One problem, all rounds in one launch: about 512 of the 1024-thread task workgroups are
    // resident at a time (two per CU).  A plan with more tasks than that serializes its tail
    // and usually has a round more than necessary; twice the task size fixes both (cart-pole
    // N=5000: 547 tasks / 4 rounds -> 265 / 3, factorization 73 -> 62 us, backward solve 42 -> 38;
    // at N=1000 the 137 tasks of the default are the better choice: 39 vs 45 us).
    // (more than ~250 tasks take two workgroups per CU, 80 KB of LDS each: there the chains-from-the-deepest-
    // child rule stays out of the leaf tasks, where it only adds fronts — tables, arena — to full levels:
    // cart-pole N=5000 58.6 us against 62.1 without the rule and 96 (not resident: two launches) with it everywhere)
    // Both rules are applied inside the build, after its first task partition (LdltOptions::single_problem_task_rules).
    lopt.single_problem_task_rules = opt.batch == 1 && std::getenv("SLPX_TASK_ENTRIES") == nullptr;
    // The reference's own choice (interior_point.hpp:340-352, sqp.hpp:238-240, newton.hpp:133-135): a system whose lower
    // triangle fills a quarter of it or more — the small problems of its unit tests, single shooting — is factored
    // dense, with Eigen::LDLT's diagonal pivoting (ldlt_dense_pivoted_factor_kernel).  SLPX_DENSE=0: the sparse plan
    // whatever the fill (the sparse kernels do not pivot: D, and with it the regularization the policy settles on, may
    // then differ from the reference's on such a system); a caller's elimination order asks for the sparse plan too.
    {
      const char* denv = std::getenv("SLPX_DENSE");
      if (m_k.reference_takes_dense(st.Ae.nnz()) && user_perm == nullptr && (denv == nullptr || denv[0] != '0') && m_k.dim <= 2048) {
        m_l = build_dense_ldlt_plan(m_k.lhs, st.n);
        m_l.dense_pivoted = denv == nullptr || denv[0] != '1';  // (SLPX_DENSE=1 keeps meaning the plain dense kernel)
        return;
      }
    }
    m_l = plan_or_dense(m_k.lhs, st.n, lopt, user_perm, &diag_has_source, opt.batch);
    if (lopt.multifrontal && !m_l.mf && !m_l.dense) {
      // the fronts were not built (a limit of their addressing, or a refused plan): the pair-list kernels run this
      // system, with THEIR tuning — chains from four columns, exact structures — not the fronts'
      LdltOptions pair = opt.ldlt;
      pair.task_entries = lopt.task_entries;
      pair.supernodal = lopt.supernodal;
      pair.single_problem_task_rules = false;
      m_l = plan_or_dense(m_k.lhs, st.n, pair, user_perm, &diag_has_source, opt.batch);
    }
  };
  m_s = build_nlp_structure(g, x, f, c_e, c_i, opt.tape, plan_linear_algebra);
  lap("= AD structure + tape compile, KKT plan, LDLT symbolic");
  reset_regularization();
  if (defer_device) return;
  device_job.get();
  finish_device();
}

NewtonSystem::~NewtonSystem() = default;

FrDevice& NewtonSystem::restoration_device() {
  if (!m_fr) m_fr = std::make_unique<FrDevice>(*m_dev);
  return *m_fr;
}

void NewtonSystem::finish_device() {
  if (m_dev) return;
  SetupLap lap;
  m_dev = std::make_unique<DeviceNlp>(m_s, m_k, m_l, m_opt.batch, m_opt.device);
  lap("= device upload + tape JIT");
}

NewtonSystem::NewtonSystem(const CscPattern& lower, int n_dec, int m_e, const NewtonOptions& opt)
    : m_opt(opt) {
  if (lower.cols != n_dec + m_e || lower.rows != lower.cols)
    throw std::runtime_error("slpx: the matrix must be square of order n + m_e");
  m_opt.batch = std::max(1, opt.batch);
  const int dim = n_dec + m_e;
  // the structure a DeviceNlp expects, with nothing in it but the sizes
  m_s.n = n_dec;
  m_s.m_e = m_e;
  m_s.m_i = 0;
  m_s.nV = 1;
  m_s.V_static_raw.assign(1, 0.0);
  m_s.V_scale_idx.assign(1, -1);
  m_s.V_is_static.assign(1, 1);
  auto empty = [](int rows, int cols) {
    CscPattern p;
    p.rows = rows;
    p.cols = cols;
    p.colptr.assign(cols + 1, 0);
    return p;
  };
  m_s.g_pat = empty(1, n_dec);
  m_s.Ae = empty(m_e, n_dec);
  m_s.Ai = empty(0, n_dec);
  m_s.Hf = empty(n_dec, n_dec);
  m_s.Hc = empty(n_dec, n_dec);
  // pattern = the caller's lower triangle plus any missing diagonal entry
  m_k.n = n_dec;
  m_k.m_e = m_e;
  m_k.m_i = 0;
  m_k.dim = dim;
  m_k.lhs.rows = m_k.lhs.cols = dim;
  m_k.lhs.colptr.assign(1, 0);
  m_user_lhs_map.assign(lower.rowidx.size(), -1);
  std::vector<uint8_t> diag_has_source(dim, 0);
  for (int c = 0; c < dim; ++c) {
    std::vector<std::pair<int32_t, int32_t>> rows;  // (row, index in the caller's arrays or -1)
    bool has_diag = false;
    for (int32_t p = lower.colptr[c]; p < lower.colptr[c + 1]; ++p) {
      const int32_t r = lower.rowidx[p];
      if (r < c || r >= dim) throw std::runtime_error("slpx: the pattern must be the lower triangle in CSC form");
      has_diag = has_diag || r == c;
      rows.emplace_back(r, p);
    }
    diag_has_source[c] = has_diag;
    if (!has_diag) rows.emplace_back(c, -1);
    std::sort(rows.begin(), rows.end());
    for (auto& [r, p] : rows) {
      if (p >= 0) m_user_lhs_map[p] = static_cast<int32_t>(m_k.lhs.rowidx.size());
      m_k.lhs.rowidx.push_back(r);
    }
    m_k.lhs.colptr.push_back(static_cast<int32_t>(m_k.lhs.rowidx.size()));
  }
  const int nnz = m_k.lhs.nnz();
  m_k.dptr.assign(nnz + 1, 0);
  m_k.pptr.assign(nnz + 1, 0);
  m_k.fast_src.assign(nnz, -1);
  m_k.g_src.assign(n_dec, -1);
  m_k.ai_rowptr.assign(1, 0);
  m_k.ae_rowptr.assign(m_e + 1, 0);
  LdltOptions lopt = opt.ldlt;
  if (opt.batch >= 16 && lopt.task_entries == LdltOptions{}.task_entries) lopt.task_entries = 1024;
  if (DeviceNlp::interleaved_for(opt.batch) && lopt.task_entries >= 1024) lopt.task_entries = opt.batch < 192 ? 512 : 384;
  if (DeviceNlp::interleaved_for(opt.batch)) lopt.supernodal = false;
  m_l = plan_or_dense(m_k.lhs, n_dec, lopt, nullptr, &diag_has_source, m_opt.batch);
  m_dev = std::make_unique<DeviceNlp>(m_s, m_k, m_l, m_opt.batch, opt.device);
  m_dev->set_scaling(std::vector<double>(m_s.n_scales(), 1.0));
  reset_regularization();
}

void NewtonSystem::reset_regularization() {
  m_prev_delta.assign(m_opt.batch, 0.0);
  m_prev_gamma.assign(m_opt.batch, 0.0);
}

// sparse_regularized_ldlt.hpp:64-152, run for every problem of the batch at once.
// Each trip through the loop is one device factorization of the still-active
// problems followed by one small stats read-back.
std::vector<FactorInfo> NewtonSystem::compute(bool solve_speculatively) {
  return compute_impl(solve_speculatively ? 1 : 0);
}

// mode 0: factorization attempts only; 1: each attempt followed by solve + backsub.
std::vector<FactorInfo> NewtonSystem::compute_impl(int mode) {
  const bool solve_speculatively = mode >= 1;
  const int B = m_opt.batch;
  m_last_twin_launches = m_last_twin_taken = 0;
  if (mode == 1 && B == 1 && m_twin_attempts && m_dev->twin_available()) return compute_twin();
  // With solve_speculatively every factorization attempt is followed at once by the
  // triangular solves and the back-substitution, BEFORE the host has read the inertia
  // counters: the device never idles through the host round trip, and in the usual case
  // (first attempt accepted) the step is complete when the counters arrive.  A rejected
  // attempt just has its solve overwritten by the next one.
  auto factor_once = [&](const std::vector<double>& d, const std::vector<double>& g,
                         const std::vector<uint8_t>& a) {
    if (solve_speculatively) {
      m_dev->factor_solve_publish(d, g, a);
      if (m_after_attempt) m_after_attempt();
    } else {
      m_dev->factor(d, g, a);
    }
  };
  std::vector<LdltStats> stats;
  // one attempt and its counters.  A chained step (DeviceNlp::sweep_full_for_step) whose sweep or step
  // kernel gave up waiting for the other reports kLdltChainFailure instead of a silent wrong step: the
  // attempt is redone — V swept again from the unchanged state, the system rebuilt — with the chain off.
  auto factor = [&](const std::vector<double>& d, const std::vector<double>& g, const std::vector<uint8_t>& a) {
    factor_once(d, g, a);
    m_dev->read_stats(stats);
    if (B == 1 && (stats[0].n_bad & kLdltChainFailure) != 0) {
      m_dev->recover_from_chain_failure();
      m_dev->build_kkt_for_step(/*with_reduce=*/true);
      factor_once(d, g, a);
      m_dev->read_stats(stats);
    }
  };
  const int n = m_s.n, m_e = m_s.m_e;
  std::vector<FactorInfo> info(B, FactorInfo::Success);
  std::vector<double> delta(B, 0.0), gamma(B, 0.0);
  std::vector<uint8_t> active(B, 1);
  m_last_factorizations = 0;
  const double eps = std::numeric_limits<double>::epsilon();

  auto inertia_ok = [&](const LdltStats& st) {
    return st.n_pos == n && st.n_neg == m_e && st.n_zero == 0;
  };
  auto min_abs = [](const LdltStats& st) {
    double d;
    std::memcpy(&d, &st.min_abs_bits, sizeof(d));
    return d;
  };

  // First attempt: unregularized (:74-87).  When the symbolic phase proved that a
  // pivot is structurally zero the attempt is known to end in NumericalIssue
  // (Eigen reports failure on an exactly-zero pivot), so it is not launched.
  std::vector<uint8_t> need_loop(B, 0);
  const bool skip_first = m_opt.skip_structurally_singular_attempt &&
                          m_l.structurally_singular_unregularized;
  if (!skip_first) {
    factor(delta, gamma, active);
    ++m_last_factorizations;
    for (int b = 0; b < B; ++b) {
      const bool success = stats[b].n_bad == 0;
      if (success && inertia_ok(stats[b]) && min_abs(stats[b]) >= 1e-4) {
        m_prev_delta[b] = 0.0;
        m_prev_gamma[b] = 0.0;
        active[b] = 0;
      } else {
        need_loop[b] = 1;
      }
    }
  } else {
    std::fill(need_loop.begin(), need_loop.end(), 1);
  }

  bool any = false;
  for (int b = 0; b < B; ++b) {
    active[b] = need_loop[b];
    if (need_loop[b]) {
      any = true;
      delta[b] = m_prev_delta[b] == 0.0 ? 1e-4 : std::max(m_prev_delta[b] / 2.0, eps);  // :95-98
      gamma[b] = m_gamma_min;                                                            // :102
    }
  }
  while (any) {
    factor(delta, gamma, active);
    ++m_last_factorizations;
    any = false;
    for (int b = 0; b < B; ++b) {
      if (!active[b]) continue;
      const LdltStats& st = stats[b];
      if (st.n_bad == 0) {
        if (inertia_ok(st)) {  // :109-113
          m_prev_delta[b] = delta[b];
          m_prev_gamma[b] = gamma[b];
          active[b] = 0;
          continue;
        } else if (st.n_zero > 0) {  // :114-126
          if (gamma[b] == 0.0) {
            gamma[b] = 1e-10;
          } else {
            delta[b] *= 10.0;
            gamma[b] *= 10.0;
          }
        } else if (st.n_neg > m_e) {  // :127-130
          delta[b] *= 10.0;
        } else if (st.n_pos > n) {  // :131-135
          gamma[b] = gamma[b] == 0.0 ? 1e-10 : gamma[b] * 10.0;
        }
      } else {  // :136-141
        delta[b] *= 10.0;
        gamma[b] = gamma[b] == 0.0 ? 1e-10 : gamma[b] * 10.0;
      }
      if (delta[b] > 1e20 || gamma[b] > 1e20) {  // :145-150
        info[b] = FactorInfo::NumericalIssue;
        m_prev_delta[b] = delta[b];
        m_prev_gamma[b] = gamma[b];
        active[b] = 0;
        continue;
      }
      any = true;
    }
  }
  return info;
}

// The policy loop of compute_impl (sparse_regularized_ldlt.hpp:64-152) for one problem, two attempts per launch:
// the attempt the loop is at and the one it makes next if this one has too many negative pivots (delta x 10,
// :127-130) — beside the unregularized first attempt, the first guess (:95-102).  The attempts are judged in the
// policy's order from their own counters, so the sequence of (delta, gamma) tried, the one accepted and the count of
// factorizations are the sequential loop's; a second attempt the policy would not have made next is ignored.
// ipm_lookahead_kernel makes the same choice on the device from the same counters: keep the two in step.
NewtonSystem::TwinLaunch NewtonSystem::twin_first_launch() const {
  const double eps = std::numeric_limits<double>::epsilon();
  const double d = m_prev_delta[0] == 0.0 ? 1e-4 : std::max(m_prev_delta[0] / 2.0, eps);  // :95-98
  const double g = m_gamma_min;                                                             // :102
  const bool skip_first = m_opt.skip_structurally_singular_attempt && m_l.structurally_singular_unregularized;
  if (!skip_first) return TwinLaunch{0.0, 0.0, d, g, 2};
  if (m_twin_expect == 3) return TwinLaunch{d, g, d, g == 0.0 ? 1e-10 : g * 10.0, 3};
  return TwinLaunch{d, g, d * 10.0, g, 1};
}

bool NewtonSystem::begin_speculative_compute(bool gated) {
  if (m_spec.valid || m_opt.batch != 1 || !m_twin_attempts || !m_dev->twin_available()) {
    m_dev->ipm_ride_disarm();
    return false;
  }
  const TwinLaunch tl = twin_first_launch();
  const DeviceNlp::LaunchBook book = m_dev->save_book();
  m_dev->build_kkt_for_step(/*with_reduce=*/false);
  m_dev->ipm_gate_next_step(gated);
  if (!m_dev->factor_solve_publish_twin(tl.d0, tl.g0, tl.d1, tl.g1, tl.mode)) {
    m_dev->ipm_gate_next_step(false);
    m_dev->ipm_ride_disarm();
    m_dev->restore_book(book);
    return false;
  }
  if (m_after_attempt) m_after_attempt();
  m_spec.valid = true;
  m_spec.have_second = true;
  m_spec.launch = tl;
  m_spec.book = book;
  return true;
}

void NewtonSystem::cancel_speculative_compute(bool launch_ran) {
  if (!m_spec.valid) return;
  m_dev->restore_book(m_spec.book, launch_ran);
  m_spec.valid = false;
}

std::vector<FactorInfo> NewtonSystem::compute_twin() {
  const int n = m_s.n, m_e = m_s.m_e;
  std::vector<FactorInfo> info(1, FactorInfo::Success);
  m_last_factorizations = 0;
  const double eps = std::numeric_limits<double>::epsilon();
  auto good = [&](const LdltStats& st) { return st.n_bad == 0 && st.n_pos == n && st.n_neg == m_e && st.n_zero == 0; };
  auto min_abs = [](const LdltStats& st) {
    double d;
    std::memcpy(&d, &st.min_abs_bits, sizeof(d));
    return d;
  };
  std::vector<LdltStats> stats;
  LdltStats first{}, second{};
  bool have_second = false;
  // one launch: (d0, g0) and, if the device can, (d1, g1) beside it
  bool first_launch = true;
  auto launch = [&](double d0, double g0, double d1, double g1, int mode) {
    // (a later launch of the loop evaluates the system from V again, inside the launch, like the first — the
    // caller's system IS the one V, s, y, z describe, set_twin_attempts — instead of two assembly launches first)
    if (!first_launch) m_dev->build_kkt_for_step(/*with_reduce=*/false);
    first_launch = false;
    auto once = [&] {
      have_second = m_dev->factor_solve_publish_twin(d0, g0, d1, g1, mode);
      if (!have_second) m_dev->factor_solve_publish({d0}, {g0}, {1});
      if (m_after_attempt) m_after_attempt();
      m_dev->read_stats(stats);
    };
    if (m_spec.valid) {
      // this launch — the policy's first of this compute — was made ahead (begin_speculative_compute) and has run
      if (m_spec.launch.d0 != d0 || m_spec.launch.g0 != g0 || m_spec.launch.d1 != d1 || m_spec.launch.g1 != g1 || m_spec.launch.mode != mode)
        throw std::runtime_error("slpx: the step enqueued ahead is not the one the regularization policy makes");
      m_spec.valid = false;
      have_second = m_spec.have_second;
      m_dev->read_stats(stats);
    } else {
      once();
    }
    // (a chained step that lost its hand-over — compute_impl's factor(): redone unchained from a fresh sweep.
    // A twin launch itself is never chained, the single-attempt fallback above can be.)
    if ((stats[0].n_bad & kLdltChainFailure) != 0) {
      m_dev->recover_from_chain_failure();
      m_dev->build_kkt_for_step(/*with_reduce=*/true);
      once();
    }
    if (have_second) ++m_last_twin_launches;
    first = stats[0];
    if (have_second) {
      second = m_dev->read_twin_stats();
      const bool first_good = good(first) && (mode != 2 || min_abs(first) >= 1e-4);
      const bool neg = first.n_neg > m_e;
      const bool second_stands = mode == 2 || (mode == 1 && neg) || (mode == 3 && !neg);
      const int kind = first_good ? 0
                       : first.n_bad != 0 ? 5
                       : first.n_zero > 0 ? 3
                       : second_stands ? (good(second) ? 1 : 2)
                                       : 4;
      ++m_twin_hist[kind];
    }
  };
  auto accept = [&](double d, double g, bool is_second) {
    m_prev_delta[0] = d;
    m_prev_gamma[0] = g;
    if (is_second) {
      m_dev->adopt_twin();
      ++m_last_twin_taken;
    }
    return info;
  };
  // the loop's answer to a failed attempt (:114-141); true: it was "too many negative pivots"
  // (returns which of the loop's answers it was: 1 too many negative pivots, 3 too many positive — the numbers of
  // the launch modes whose second attempt stands for that answer, IpmTwin::mode —, 0 anything else)
  auto advance = [&](const LdltStats& st, double& d, double& g) {
    int answer = 0;
    if (st.n_bad == 0) {
      if (st.n_zero > 0) {
        if (g == 0.0) {
          g = 1e-10;
        } else {
          d *= 10.0;
          g *= 10.0;
        }
      } else if (st.n_neg > m_e) {
        d *= 10.0;
        answer = 1;
      } else if (st.n_pos > n) {
        g = g == 0.0 ? 1e-10 : g * 10.0;
        answer = 3;
      }
    } else {
      d *= 10.0;
      g = g == 0.0 ? 1e-10 : g * 10.0;
    }
    return answer;
  };
  auto gave_up = [&](double d, double g) {  // :145-150
    if (!(d > 1e20 || g > 1e20)) return false;
    info[0] = FactorInfo::NumericalIssue;
    m_prev_delta[0] = d;
    m_prev_gamma[0] = g;
    return true;
  };

  double d = m_prev_delta[0] == 0.0 ? 1e-4 : std::max(m_prev_delta[0] / 2.0, eps);  // :95-98
  double g = m_gamma_min;                                                             // :102
  bool second_is_current = false;  // `second` holds the attempt at (d, g)
  const bool skip_first = m_opt.skip_structurally_singular_attempt && m_l.structurally_singular_unregularized;
  if (!skip_first) {  // :74-87
    launch(0.0, 0.0, d, g, 2);
    ++m_last_factorizations;
    if (good(first) && min_abs(first) >= 1e-4) return accept(0.0, 0.0, false);
    second_is_current = have_second;
  }
  // Which answer the second attempt of a launch stands for: the one the loop's first attempt drew in the LAST
  // compute() (a phase of the solve that needs a larger gamma needs it iteration after iteration — the loop starts
  // from gamma_min every time, :102), too many negative pivots otherwise and for the later launches of a loop.
  int expect = m_twin_expect;
  bool loop_first = true;
  while (true) {
    if (!second_is_current) {
      if (expect == 3) launch(d, g, d, g == 0.0 ? 1e-10 : g * 10.0, 3);
      else launch(d, g, d * 10.0, g, 1);
      ++m_last_factorizations;
      if (good(first)) {
        if (loop_first) m_twin_expect = 1;
        return accept(d, g, false);
      }
      const int answer = advance(first, d, g);
      if (loop_first && answer != 0) m_twin_expect = answer;
      loop_first = false;
      if (gave_up(d, g)) return info;
      second_is_current = have_second && answer == expect;  // (d, g) is now what the second attempt was made with
      expect = 1;
      if (!second_is_current) continue;
    }
    second_is_current = false;
    ++m_last_factorizations;
    if (good(second)) return accept(d, g, true);
    advance(second, d, g);
    if (gave_up(d, g)) return info;
  }
}

std::vector<FactorInfo> NewtonSystem::compute_hooked(const AttemptHooks& hooks) {
  if (m_opt.batch != 1) throw std::runtime_error("slpx: compute_hooked handles one problem");
  const int n = m_s.n, m_e = m_s.m_e;
  std::vector<FactorInfo> info(1, FactorInfo::Success);
  m_last_factorizations = 0;
  m_last_twin_launches = m_last_twin_taken = 0;
  m_hooked_chain_valid = false;
  const double eps = std::numeric_limits<double>::epsilon();
  std::vector<LdltStats> stats;
  LdltStats first{}, second{};
  bool have_second = false, chain_behind_first = false;
  const bool twin = static_cast<bool>(hooks.prepare_second) && m_dev->twin_available();
  // one launch: (d0, g0) and, where the device can, (d1, g1) beside it
  auto launch = [&](double d0, double g0, double d1, double g1, int mode) {
    have_second = false;
    if (twin) {
      const double *lhs2 = nullptr, *rhs2 = nullptr;
      if (hooks.prepare_pair) {
        hooks.prepare_pair(d0, g0, d1, g1, &lhs2, &rhs2);
      } else {
        hooks.prepare(d0, g0);
        hooks.prepare_second(d1, g1, &lhs2, &rhs2);
      }
      have_second = m_dev->factor_solve_publish_twin_written(d0, g0, d1, g1, mode, lhs2, rhs2);
    } else {
      hooks.prepare(d0, g0);
    }
    if (!have_second) m_dev->factor_solve_publish({d0}, {g0}, {1});
    // (`after` behind the FIRST launch only: a caller that expects its first attempt to be taken; the launches of a
    // ladder would each drag a chain nobody reads)
    chain_behind_first = false;
    if (hooks.after && m_last_factorizations == 0) {
      hooks.after(d0, g0);
      chain_behind_first = true;
    }
    m_dev->read_stats(stats);
    first = stats[0];
    if (have_second) {
      second = m_dev->read_twin_stats();
      ++m_last_twin_launches;
    }
  };
  auto good = [&](const LdltStats& st) { return st.n_bad == 0 && st.n_pos == n && st.n_neg == m_e && st.n_zero == 0; };
  auto min_abs = [](const LdltStats& st) {
    double d;
    std::memcpy(&d, &st.min_abs_bits, sizeof(d));
    return d;
  };
  auto accept = [&](double d, double g, bool is_second) {
    m_prev_delta[0] = d;
    m_prev_gamma[0] = g;
    if (is_second) {
      m_dev->adopt_twin();
      ++m_last_twin_taken;
    }
    m_hooked_chain_valid = chain_behind_first && !is_second;
    return info;
  };
  // the loop's answer to a failed attempt (:114-141); 1: it was "too many negative pivots" — what a launch's second
  // attempt (delta x 10) stands for
  auto advance = [&](const LdltStats& st, double& d, double& g) {
    int answer = 0;
    if (st.n_bad == 0) {
      if (st.n_zero > 0) {
        if (g == 0.0) {
          g = 1e-10;
        } else {
          d *= 10.0;
          g *= 10.0;
        }
      } else if (st.n_neg > m_e) {
        d *= 10.0;
        answer = 1;
      } else if (st.n_pos > n) {
        g = g == 0.0 ? 1e-10 : g * 10.0;
      }
    } else {
      d *= 10.0;
      g = g == 0.0 ? 1e-10 : g * 10.0;
    }
    return answer;
  };
  auto gave_up = [&](double d, double g) {  // :145-150
    if (!(d > 1e20 || g > 1e20)) return false;
    info[0] = FactorInfo::NumericalIssue;
    m_prev_delta[0] = d;
    m_prev_gamma[0] = g;
    return true;
  };
  double d = m_prev_delta[0] == 0.0 ? 1e-4 : std::max(m_prev_delta[0] / 2.0, eps);  // :95-98
  double g = m_gamma_min;                                                             // :102
  launch(0.0, 0.0, d, g, 2);  // :74-87 — beside it the loop's first guess
  ++m_last_factorizations;
  if (good(first) && min_abs(first) >= 1e-4 && (!hooks.eliminated_min_pivot || hooks.eliminated_min_pivot() >= 1e-4))
    return accept(0.0, 0.0, false);
  bool second_is_current = have_second;  // `second` holds the attempt at (d, g)
  while (true) {
    if (!second_is_current) {
      launch(d, g, d * 10.0, g, 1);
      ++m_last_factorizations;
      if (good(first)) return accept(d, g, false);
      const int answer = advance(first, d, g);
      if (gave_up(d, g)) return info;
      second_is_current = have_second && answer == 1;  // (d, g) is now what the second attempt was made with
      if (!second_is_current) continue;
    }
    second_is_current = false;
    ++m_last_factorizations;
    if (good(second)) return accept(d, g, true);
    advance(second, d, g);
    if (gave_up(d, g)) return info;
  }
}

bool NewtonSystem::factor_unregularized() {
  const int B = m_opt.batch;
  std::vector<double> zero(B, 0.0);
  std::vector<uint8_t> active(B, 1);
  m_dev->factor(zero, zero, active);
  std::vector<LdltStats> stats;
  m_dev->read_stats(stats);
  ++m_last_factorizations;
  for (int b = 0; b < B; ++b)
    if (stats[b].n_bad != 0 || stats[b].n_pos != m_s.n || stats[b].n_neg != m_s.m_e || stats[b].n_zero != 0)
      return false;
  return true;
}

std::vector<FactorInfo> NewtonSystem::newton_step(bool refresh_ad) {
  // SLPX_HOST_TIMING=1: where the host's time per step goes (printed every 1000 steps)
  static const bool timing = std::getenv("SLPX_HOST_TIMING") != nullptr;
  if (!timing) {
    if (refresh_ad) m_dev->sweep_full_for_step();
    m_dev->build_kkt_for_step(/*with_reduce=*/refresh_ad);
    return compute(/*solve_speculatively=*/true);
  }
  using clk = std::chrono::steady_clock;
  static double t_sweep = 0, t_rest = 0;
  static long n = 0;
  const auto t0 = clk::now();
  if (refresh_ad) m_dev->sweep_full_for_step();
  const auto t1 = clk::now();
  m_dev->build_kkt_for_step(/*with_reduce=*/refresh_ad);
  auto res = compute(/*solve_speculatively=*/true);
  const auto t2 = clk::now();
  t_sweep += std::chrono::duration<double, std::micro>(t1 - t0).count();
  t_rest += std::chrono::duration<double, std::micro>(t2 - t1).count();
  if (++n % 1000 == 0) {
    std::fprintf(stderr, "slpx host timing: sweep launch %.2f us, launch + wait for the verdict %.2f us per step\n",
                 t_sweep / 1000, t_rest / 1000);
    t_sweep = t_rest = 0;
  }
  return res;
}

}  // namespace slpx
