// Device-side state of one compiled NLP on one MI355X: uploaded plans, per-batch
// value buffers, and launchers for the hot-path kernels (kernels.hip).
//
// Batch-major layout everywhere: buffer[b * stride + i], b = problem in the batch.
// All problems of a batch share one structure (tape, KKT plan, symbolic LDLᵀ) and
// differ only in values — the reference's analogue is multistart's one-thread-per-
// solve fan-out (include/sleipnir/optimization/multistart.hpp:52-62).
#pragma once

#include <chrono>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kkt_plan.hpp"
#include "ldlt_symbolic.hpp"
#include "nlp.hpp"

namespace slpx {

#define SLPX_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t err__ = (expr);                                                            \
    if (err__ != hipSuccess)                                                              \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(err__) +   \
                               " at " #expr);                                             \
  } while (0)

// The read-only plan arrays of a system (a hundred of them: tape tables, KKT maps, LDLT lists, task images) in a
// few large device allocations filled by ONE copy each, instead of a hipMalloc and a synchronous hipMemcpy per array
// (2-3 ms of a 7 ms upload at cart-pole N=1000): while an arena is the thread's current one (DeviceArena::Scope,
// DeviceNlp's constructor), DevBuf::upload() places its data in the arena — the device pointer is valid at once, the
// bytes arrive with commit(), before anything is launched.
class DeviceArena {
 public:
  DeviceArena() = default;
  DeviceArena(const DeviceArena&) = delete;
  DeviceArena& operator=(const DeviceArena&) = delete;
  ~DeviceArena() {
    for (Chunk& c : m_chunks)
      if (c.dev) (void)hipFree(c.dev);
  }
  void* place(const void* src, size_t bytes) {
    constexpr size_t kAlign = 256, kChunk = 2u << 20, kSlack = 128u << 10;  // (slack: kernels that request a padded round past an array's end)
    if (m_chunks.empty() || m_chunks.back().used + bytes + kSlack > m_chunks.back().cap) {
      Chunk c;
      c.cap = std::max(kChunk, bytes + 2 * kSlack);
      if (hipMalloc(reinterpret_cast<void**>(&c.dev), c.cap) != hipSuccess) throw std::runtime_error("slpx: hipMalloc of a plan arena failed");
      m_chunks.push_back(std::move(c));
    }
    Chunk& c = m_chunks.back();
    const size_t at = c.used;
    if (c.host.capacity() < c.cap) c.host.reserve(c.cap);
    if (c.host.size() < at + bytes) c.host.resize(at + bytes);
    std::memcpy(c.host.data() + at, src, bytes);
    c.used = (at + bytes + kAlign - 1) / kAlign * kAlign;
    return c.dev + at;
  }
  void commit() {
    for (Chunk& c : m_chunks) {
      if (c.host.size() > c.committed) {
        if (hipMemcpy(c.dev + c.committed, c.host.data() + c.committed, c.host.size() - c.committed, hipMemcpyHostToDevice) != hipSuccess)
          throw std::runtime_error("slpx: upload of a plan arena failed");
        c.committed = c.host.size();
      }
      if (&c != &m_chunks.back()) std::vector<char>().swap(c.host);
    }
    if (!m_chunks.empty()) {  // (the open chunk keeps no mirror either: later placements start a new one)
      std::vector<char>().swap(m_chunks.back().host);
      m_chunks.back().cap = m_chunks.back().used;
    }
  }
  static DeviceArena*& current() {
    static thread_local DeviceArena* arena = nullptr;
    return arena;
  }
  struct Scope {
    DeviceArena* prev;
    explicit Scope(DeviceArena* a) : prev(current()) { current() = a; }
    ~Scope() { current() = prev; }
  };

 private:
  struct Chunk {
    char* dev = nullptr;
    std::vector<char> host;  // mirror of [0, used) until commit()
    size_t used = 0, committed = 0, cap = 0;
  };
  std::vector<Chunk> m_chunks;
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = true;  // false: a slice of a DeviceArena
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
    std::swap(owned, o.owned);
  }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    n = 0;
    owned = true;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) SLPX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
  }
  void upload(const std::vector<T>& h) {
    // (the small arrays — most of them — share an arena; a big one gains nothing from a second host copy)
    if (DeviceArena* arena = DeviceArena::current(); arena != nullptr && !h.empty() && h.size() * sizeof(T) <= (64u << 10)) {
      release();
      p = static_cast<T*>(arena->place(h.data(), h.size() * sizeof(T)));
      n = h.size();
      owned = false;
      return;
    }
    alloc(h.size());
    if (!h.empty()) SLPX_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  void zero(hipStream_t s = nullptr) {
    if (n) SLPX_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

struct TapeDevice {
  DevBuf<TapeTask> tasks;
  DevBuf<uint32_t> small_list, large_list, global_list;
  DevBuf<uint32_t> leaf_src, node_rec, lvl_ptr, slot_edge_ptr, slvl_ptr, vout_src, vout_dst,
      jout_slot, jout_dst;
  DevBuf<int32_t> vout_scale, jout_scale;
  DevBuf<double> consts;
  DevBuf<TapeEdge> edges;
  DevBuf<uint16_t> node_rec16, slot_edge_ptr16, edges16;
  uint32_t n_small = 0, n_large = 0, n_global = 0;
  uint32_t small_lds = 0, large_lds = 0;
  uint64_t scratch_doubles = 0;
  bool basic_ops = false;
  // Tasks served by the run-time generated lane-per-task kernel (tape_jit.hpp); they are
  // NOT in the small/large lists above.  One launch covers every body: `tmpl_table[mode]`
  // (mode 0 = values only, 1 = with adjoints) holds (first block, instances, instance
  // offset, row-group mode) per body, `tmpl_blocks[mode]` the grid size.
  hipFunction_t tmpl_fn = nullptr;
  uint32_t tmpl_threads = 64;  // threads of a workgroup of tmpl_fn (TapeJitResult::block_threads)
  std::vector<unsigned char> tmpl_params;  // the model's numbers a generic code object reads (TemplateParams::blob)
  DevBuf<uint64_t> tmpl_params_dev;        // ... on the device: the kernel's last argument points here
  hipModule_t tmpl_mod = nullptr;
  uint32_t n_bodies = 0;
  DevBuf<uint32_t> tmpl_inst;  // per body: leaf bindings + output destinations, transposed (kernels.hip)
  DevBuf<uint32_t> tmpl_table[2];
  uint32_t tmpl_blocks[2] = {0, 0};
  uint32_t n_templated_tasks = 0;
  bool tmpl_wide_for_chain = false;  // 256-thread workgroups only because the sweep runs beside the step kernel (tape_jit.cpp)
  double jit_seconds = 0.0;
  void upload(const TapeProgram& p, int batch, uint32_t n_unscaled_inputs, int chain_mode = 0);
  TapeDev view() const;
};

struct LdltDev {
  const LdltTask* tasks;
  const int32_t* ent_src;
  const uint8_t* ent_flags;
  const uint16_t* ent_col;
  const uint32_t* ent_out;
  const uint32_t* ent_pair_ptr;
  const uint32_t* ent_contrib_ptr;
  const uint32_t* contrib_idx;
  const uint32_t* ext_dst;
  const uint32_t* lvl_ptr;
  const LdltPair* pairs;
  const uint32_t* col_perm;
  const uint32_t* col_lvl_ptr;
  const uint32_t* fwd_ptr;
  const uint32_t* fwd_contrib_ptr;
  const uint32_t* scontrib_idx;
  const LdltSolveItem* fwd_items;
  const uint32_t* sext_ptr;
  const uint32_t* sext_dst;
  const LdltSolveItem* sext_items;
  const uint32_t* bwd_ptr;
  const LdltSolveItem* bwd_items;
  const int32_t* perm;
  const uint32_t* round_ptr;  // n_rounds + 1, into tasks
  int n_rounds;
  // supernodes (ldlt_symbolic.hpp: LdltSn)
  const LdltSn* sn_desc;
  const uint32_t* sn_lvl_ptr;
  const uint32_t* col_sn;
  // the hot loops' own copies (one LDS word instead of two per level / per column):
  const uint32_t* lvl_pack;      // lvl_ptr | sn_lvl_ptr << 16
  const uint32_t* col_lvl_pack;  // col_lvl_ptr | sn_lvl_ptr << 16
  const uint2* bwd_range;        // per column {first item below its own chain, end}; shares col_off
  uint32_t clock_task;           // the task whose phase clocks are recorded (slpx_debug_ldlt_clocks); 0xffffffff: none
};

// n_bad bit of a chained step whose sweep / step kernel gave up waiting for the other (bounded spins)
constexpr int32_t kLdltChainFailure = 1 << 30;

struct LdltStats {   // one per batch item, written by the factor kernels
  int32_t n_pos, n_neg, n_zero, n_bad;  // n_bad: exactly-zero or non-finite pivots (+ 2^20 per hand-over that timed out; bit 30: kLdltChainFailure)
  unsigned long long min_abs_bits;      // bit pattern of min |D| (non-negative doubles order like integers)
};

struct KktDev {
  int n, m_e, m_i, dim, nnz_lhs;
  const int32_t* dptr;
  const int32_t* dsrc;
  const int32_t* pptr;
  const int32_t* pa;
  const int32_t* pb;
  const int32_t* pr;
  const int32_t* g_src;
  const int32_t* ae_colptr;
  const int32_t* ae_rowidx;
  const int32_t* ai_colptr;
  const int32_t* ai_rowidx;
  const int32_t* ai_rowptr;
  const int32_t* ai_col;
  const int32_t* ai_src;
  const int32_t* fast_src;
  int off_f, off_ce, off_ci, off_g, off_Ae, off_Ai;
};

// Work that rides in another kernel's launch (one problem, single-launch factorization).  Every
// kernel boundary of a Newton step costs 1-2 us of idle GPU plus the consumer's start-up, and an
// assembly pass in front of the factorization is a chain of three dependent trips to memory
// with the result written out only to be gathered again; so
//  - KktFuse: the factorization's tasks evaluate the lhs / rhs entries they own straight from
//    the AD sweep's V (kkt_kernels.h: one more dependent load in their gather, no assembly pass,
//    no lhs / rhs in memory unless somebody asks: DeviceNlp::materialize_kkt); the tape's
//    separable sums, which only feed f, ride as n_blocks extra workgroups nobody waits for;
//  - BacksubFuse: the step's back-substitution (p_s, p_z; interior_point.hpp:479-480) is done by
//    the backward solve's tasks themselves.  The columns of a row of A_i are pairwise adjacent in
//    the lhs (A_i^T Sigma A_i), so they lie on one root path of the elimination tree: the task
//    that owns the DEEPEST of them has the others among its own columns or its ancestors', final
//    when it finishes.  Its rows (BsRow / BsTerm) are staged with its plan, their A_i, s, z, c_i
//    values fetched with the factor's values at the start, the ancestors' p with the fold-in —
//    nothing of it is left on the critical path.  The last workgroup of the launch also hands
//    the factorization's inertia counters to the host — at the START of the launch: the host
//    learns the verdict of the attempt while the solve is still running and has the next
//    launch queued by the time it ends.
// One addend of an lhs / rhs entry that is a sum (kkt_kernels.h: kkt_terms_value); twelve bytes,
// staged into LDS with the task's plan.  kind = b >> 28, row = b & 0x0fffffff:
//   0  direct += V[a]                 1  g = V[a]                  2  aey += V[a] * y[row]
//   3  ait += V[a] * (-(z/s)[row] * V[c] + mu / s[row] + z[row])   4  prod += (V[a] * (z/s)[row]) * V[c]
struct KktTerm {
  int32_t a, b, c;
};
struct KktFuse {
  int inline_kkt = 0;
  int n_blocks = 0;  // separable sums riding along
  const double* V = nullptr;
  const double* s = nullptr;
  const double* y = nullptr;
  const double* z = nullptr;
  const double* mu = nullptr;
  // per entry of the factorization plan (same layout as LdltDev::ent_src): the V slot of a plain
  // copy (bit 30: negated), -1 a structural zero, <= -2 a sum: -(2 + (first term | count << 20)),
  // first term relative to the task's block
  const int32_t* ent_vsrc = nullptr;
  const uint4* terms = nullptr;        // KktTerm[], every task's block padded to 16 bytes
  const uint2* task_terms = nullptr;   // per task {offset of its block in uint4, its length in uint4}
  double* Vw = nullptr;
  double* store_lhs = nullptr;  // SLPX_FUSE_KKT_STORE=1 (tests): also write the evaluated system to memory
  double* store_rhs = nullptr;
  const NlpStructure::SumReduce* red = nullptr;
  const double* scales = nullptr;
};
struct BsRow {
  int32_t r;       // row of A_i
  uint32_t terms;  // first term (relative to the task's block) | number of terms << 20
};
struct BsTerm {
  int32_t a;     // V slot of the A_i value
  uint32_t ref;  // the task's own column (local index) or 0x80000000 | permuted column of an ancestor
};
struct BacksubFuse {
  int on = 0;
  const double* V = nullptr;
  const double* s = nullptr;
  const double* z = nullptr;
  const double* mu = nullptr;
  double* ps = nullptr;
  double* pz = nullptr;
  int off_ci = 0;
  const uint4* plan = nullptr;       // per task: BsRow[n_rows] (padded to 16 bytes), then BsTerm[]
  const uint4* task_plan = nullptr;  // per task {offset of its block in uint4, length in uint4, rows, uint4 of rows}
  const LdltStats* stats_src = nullptr;
  LdltStats* stats_host = nullptr;
  unsigned long long* seq_dev = nullptr;
  volatile unsigned long long* seq_host = nullptr;
};

struct IpmDirOut;
// the second attempt of a twin launch (ldlt_mf_twin_kernel), for the launch that takes the direction
struct IpmTwin {
  int mode = 0;  // 0: one attempt; 1: the policy loop's attempt and its delta x 10; 2: the unregularized attempt and the first guess; 3: the loop's attempt and its gamma x 10
  const double *p = nullptr, *ps = nullptr, *pz = nullptr;
  const LdltStats* stats = nullptr;
};
// ipm_lookahead_kernel (ipm_kernels.h): step sizes of the direction and the whole look-ahead iterate
struct IpmLookaheadArgs {
  int n = 0, m_e = 0, m_i = 0;
  const int32_t* g_src = nullptr;
  const double *V = nullptr, *in = nullptr, *s = nullptr, *y = nullptr, *z = nullptr, *p = nullptr, *ps = nullptr, *pz = nullptr,
               *mu = nullptr;
  double tau = 0.0;
  double *in_t = nullptr, *s_t = nullptr, *y_t = nullptr, *z_t = nullptr, *alpha_dev = nullptr;
  IpmDirOut* out = nullptr;
  const LdltStats* stats = nullptr;
  IpmTwin tw;
};

// Scalars the interior-point iteration kernels (ipm_kernels.h) hand to the host; the
// three blocks live side by side in pinned host memory and are written by the kernels.
struct IpmDirOut {
  double alpha_max, alpha_z, D_phi;
};
struct IpmTrialOut {
  double f, viol, logsum, finite;
};
struct IpmErrOut {
  // un-scaled quantities (kkt_error.hpp:216-251) for the termination test ...
  double dual_inf_u, sz_max_u, ce_inf_u, cis_inf_u, y1_u, z1_u;
  // ... and the scaled ones for the barrier-parameter test: ‖g − A_eᵀy − A_iᵀz‖∞,
  // min / max of s∘z, ‖c_e‖∞, ‖c_i − s‖∞, ‖y‖₁, ‖z‖₁
  double dual_inf, sz_min, sz_max, ce_inf, cis_inf, y1, z1;
  // filter entry of the current iterate: f, ‖c_e‖₁ + ‖c_i − s‖₁, Σ ln s
  double f, viol, logsum;
  // local infeasibility: ‖A_eᵀc_e‖₂², ‖c_e‖₂², ‖A_iᵀc_i⁻‖₂², ‖c_i⁻‖₂²
  double aetce_sq, ce_sq, aitcp_sq, cp_sq;
  double x_inf, s_inf, finite, ci_all_pos;
};
struct IpmHost {
  IpmDirOut dir;
  IpmTrialOut trial;
  IpmErrOut err;
  IpmErrOut err_ahead;  // the same quantities at the look-ahead iterate (ipm_lookahead)
  double go;            // a deciding error launch (ipm_errors_deciding): 1.0 = the device took the iteration's decisions
  // A riding error launch (IpmErrFinish::ride_verdict) hands over its scalars WITHOUT a sequence number: `go` = +- its
  // ticket is the host's word that err_ahead is in — and writes from the device to host memory may pass each other on
  // the way (seen: one solve in 1 500 read a half-arrived err_ahead).  `check` = the exclusive-or of err_ahead's 24
  // words and of `go`: the host takes the hand-over when what it reads adds up (DeviceNlp::ipm_ride_wait).
  unsigned long long check;
};

struct StepTimings {
  float sweep = 0, assemble = 0, rhs = 0, factor = 0, solve = 0, backsub = 0, total = 0;
};

// The stream of a DeviceNlp, remembering whether anything was handed it since the last chained Newton
// step (DeviceNlp::sweep_full_for_step): every use as a hipStream_t marks it.
struct TrackedStream {
  hipStream_t s = nullptr;
  mutable bool touched = true;
  // a chained sweep is in flight on `tape` and the step kernel that waits for it has not been launched:
  // whatever else is handed this stream first has to come after the sweep
  hipStream_t tape = nullptr;
  hipEvent_t ev = nullptr;
  mutable bool tape_pending = false;
  operator hipStream_t() const {
    touched = true;
    if (tape_pending) {
      tape_pending = false;
      (void)hipEventRecord(ev, tape);
      (void)hipStreamWaitEvent(s, ev, 0);
    }
    return s;
  }
  TrackedStream& operator=(hipStream_t v) {
    s = v;
    touched = true;
    return *this;
  }
  hipStream_t raw() const { return s; }
};

// Spin on a word a kernel publishes into pinned host memory.  The stream itself is consulted only after two
// milliseconds without progress (so that a failed launch cannot hang the host): a hipStreamQuery puts a marker packet
// behind whatever is queued, and the device spends ~5 us on it in front of the NEXT launch — measured as the gap
// before every step kernel while the loops asked every few microseconds (profiles/r06_gap_probe.txt).
template <class Done>
inline void spin_on_published(Done done, hipStream_t stream, const char* what) {
  unsigned spins = 0;
  bool timing = false;
  std::chrono::steady_clock::time_point since;
  while (!done()) {
    if ((++spins & 0x3fffu) != 0) continue;
    const auto now = std::chrono::steady_clock::now();
    if (!timing) {
      timing = true;
      since = now;
    } else if (now - since > std::chrono::milliseconds(2)) {
      since = now;
      const hipError_t st = hipStreamQuery(stream);
      if (st != hipErrorNotReady) {
        if (st != hipSuccess) throw std::runtime_error(std::string("slpx: ") + hipGetErrorString(st));
        if (!done()) throw std::runtime_error(what);
      }
    }
  }
}

class DeviceNlp {
 public:
  DeviceNlp(const NlpStructure& s, const KktPlan& k, const LdltPlan& l, int batch, int device);
  ~DeviceNlp();

  int batch() const { return m_batch; }
  void set_stream(hipStream_t s) { m_stream = s; }
  hipStream_t stream() const { return m_stream; }

  // scaling vectors: scales = [d_f, d_ce.., d_ci..]; applied to tape outputs, and
  // d_ce / d_ci also scale the dual inputs (problem.hpp:631-634).  Re-derives the
  // static part of V.
  void set_scaling(const std::vector<double>& scales);

  // state upload/download (host pointers, batch-major)
  void upload_x(const double* x);
  void upload_duals(const double* s, const double* y, const double* z);
  void upload_mu(const double* mu);  // one per batch item
  void download_V(double* V);
  void download(const double* dev, double* host, size_t count);

  // hot path (all asynchronous on stream())
  // f, c_e, c_i, g, A_e, A_i, H_f, H_c -> V.  with_reduce = false leaves the separable-sum
  // reductions (they only feed f) to the build_kkt(true) that must follow.
  void sweep_full(bool with_reduce = true);
  // The sweep of a Newton step whose factor_solve_publish() follows at once (no reductions: they ride in
  // that launch).  Where the step is the one-launch multifrontal kernel and this step follows another
  // directly (nothing else was handed the main stream in between), the sweep goes to a stream of its own
  // and the two kernels order themselves through words in memory (m_chain below): the step kernel is
  // dispatched and stages its plan WHILE the sweep runs, and the sweep of the next step starts the moment
  // this step's kernel is through — no launch boundary on either side.  DESIGN.md §4, "chained steps".
  void sweep_full_for_step();
  void recover_from_chain_failure();  // after a step whose counters carry kLdltChainFailure
  int chain_failures() const { return m_chain_failures; }
  // test hook: the next chained sweep is launched as if the step kernel before it never signalled
  void debug_break_next_chain() { m_debug_break_chain = true; }
  void sweep_values();  // f, c_e, c_i only                   -> V
  void assemble();      // V, s, z -> lhs
  void build_kkt(bool with_reduce);  // assemble() + build_rhs() [+ reductions] as one launch
  void build_kkt_for_step(bool with_reduce);  // same, inside the launch of the factor() that must follow
  // Least-squares multiplier estimate system on the same pattern:
  // lhs = [[I + A_i^T S^-2 A_i, A_e^T],[A_e, 0]] (util/lagrange_multiplier_estimate.hpp:56-133
  // with d, t eliminated; see ipm.cpp)
  void assemble_lsq();  // V, s -> lhs
  void build_rhs();     // V, s, y, z, mu -> rhs
  // Re-reads the values of the tapes' parameter leaves (free Variables that are not
  // decision variables) from the graph and refreshes their constant slots.
  void refresh_params(const Graph& g);
  void debug_tape_clocks(unsigned long long* out16);
  // SLPX_TAPE_JIT_CLOCKS=1: 8 wall clocks (tape_jit.cpp) for each of the first `blocks` workgroups of the
  // generated kernel's last launch; returns the number of template workgroups, -1 if absent
  int debug_tmpl_clocks(unsigned long long* out, int blocks);
  // returns the clocks recorded so far, then selects the round that records next
  void debug_ldlt_clocks(unsigned int next_round, unsigned long long* out24);  // phase clocks of workgroup 0 (100 MHz ticks)
  // lhs -> L, D, stats; per-problem (δ, γ), problems with active[b] == 0 are skipped
  void factor(const std::vector<double>& delta, const std::vector<double>& gamma,
              const std::vector<uint8_t>& active);
  void read_stats(std::vector<LdltStats>& out);     // synchronizes
  void solve();                                     // rhs -> p (dim per batch item)
  void solve_after_factor();                        // p for the rhs that was in place at factor()
  bool step_is_one_launch() const { return m_fuse_solve && m_fuse_kkt; }
  bool step_is_multifrontal() const { return m_mf; }
  bool factorization_is_dense() const { return m_dense; }
  void solve_backsub_publish();                     // solve_after_factor() + backsub_publish(), one launch where possible
  // factor() + solve_backsub_publish(): ONE launch where every task's workgroup fits on the device at once
  void factor_solve_publish(const std::vector<double>& delta, const std::vector<double>& gamma,
                            const std::vector<uint8_t>& active);
  void refine_solution(int iters);                  // iterative refinement of the last solve() against the lhs
  void backsub();                                   // p -> p_x, p_y, p_s, p_z
  void backsub_and_publish(const LdltStats* stats_src);
  void backsub_publish();
  void materialize_factor();                // batch-interleaved mode: refresh the batch-major L, D copies
  static bool interleaved_for(int batch);   // batches this large factor with one lane per problem
  // Twin attempt (ldlt_mf_twin_kernel): the policy loop's attempt (delta0, gamma0) and the one it would make
  // next (delta1, gamma1) in ONE launch; `mode` as IpmTwin::mode.  false: not possible now (the caller makes a
  // single attempt).  read_stats() then has the first attempt's counters, read_twin_stats() the second's;
  // adopt_twin() makes the second attempt's factor, direction and counters the system's.
  bool twin_available();
  bool factor_solve_publish_twin(double delta0, double gamma0, double delta1, double gamma1, int mode);
  // ... for a caller that wrote BOTH attempts' systems itself (restoration.hpp): the first in lhs_raw() / rhs_raw(), the
  // second in (lhs2, rhs2)
  bool factor_solve_publish_twin_written(double delta0, double gamma0, double delta1, double gamma1, int mode, const double* lhs2,
                                         const double* rhs2);
  LdltStats read_twin_stats() const { return m_h_stats[1]; }
  void adopt_twin();

  // ---- interior-point iteration on the device (ipm_kernels.h; one problem) ----
  // All asynchronous on stream(); results arrive in ipm_host() after wait().
  void ipm_enable();                          // allocates the trial / correction buffers
  // [d_f | d_ce | d_ci] the termination test un-scales with (kkt_error.hpp:216-251); not
  // necessarily the scaling the tapes apply (feasibility restoration scales in its model)
  void ipm_set_error_scaling(const std::vector<double>& scales);
  void ipm_direction(double tau);             // step sizes, D_phi; trial x = x + alpha_max p_x
  void ipm_trial_point(double alpha);         // trial x = x + alpha p_x
  void sweep_values_trial();                  // f, c_e, c_i at the trial x -> trial V
  void ipm_trial_metrics(double alpha, bool s_from_ci);  // alpha < 0: the device's alpha_max
  void ipm_commit(double alpha, double alpha_z, bool s_from_ci);
  void ipm_errors(bool check_all_V, bool sums_ride = false, bool ahead = false);  // -> IpmHost::err (one launch; the last workgroup folds)
  // Look-ahead iteration: the iterate the FULL step alpha_max would give — x, s, y, z with the z reset of
  // interior_point.hpp:797-801 — in a second set of buffers, swept (values AND derivatives) and reduced to
  // IpmHost::err_ahead speculatively; when the filter accepts that trial point the buffers change roles
  // (ipm_accept_lookahead: no launch) and the iteration is complete after ONE host round trip.
  void ipm_lookahead(double tau);             // step sizes, D_phi -> IpmHost::dir; the look-ahead iterate
  // the full tape at it, into the look-ahead V (sums ride in ipm_errors).  with_reduce: the sweep finishes its sums
  // itself; skippable = false: not part of a chain ipm_lookahead flags as void (restoration.hpp makes its own iterate)
  void sweep_full_lookahead(bool with_reduce = false, bool skippable = true);
  void ipm_accept_lookahead();                // the look-ahead iterate and its V become the current ones
  void ipm_soc_accumulate(double alpha, bool first, bool s_from_ci);
  void ipm_soc_rhs();                         // -> rhs
  void ipm_soc_backsub();                     // p -> p_s, p_z with the corrected c_i - s
  void ipm_save_direction();
  void ipm_restore_direction();
  void wait();                                // busy-polls the stream
  void wait_published();                      // after ipm_trial_metrics / ipm_errors: their sequence number
  const IpmHost& ipm_host() const { return *m_ipm_host; }
  // ---- the pipelined common iteration (ipm.cpp: ipm_core_resident; ipm_decide.h) ----
  // Two sets of the scalar outputs: the chain of iteration k+1 is enqueued before the host has read iteration k's.
  // ipm_set_slot: where the kernels enqueued from now on leave theirs.
  void ipm_set_slot(int slot) { m_ipm_slot = slot & 1; }
  const IpmHost& ipm_host_slot(int slot) const { return m_ipm_host[slot & 1]; }
  // the iteration's state for the decision on the device (host -> device, in stream order), and what the device left
  void ipm_pipeline_upload(const struct IpmCtl& ctl);
  const struct IpmCtl& ipm_pipeline_fetch();  // device -> host (synchronizes; the pipeline is drained when the host asks)
  // the NEXT step launch (one launch) waits for the decision of the error launch in front of it
  void ipm_gate_next_step(bool on) { m_gate_next_step = on; }
  // The error launch of the iteration whose look-ahead iterate was just made current (ipm_accept_lookahead) RIDES in
  // the next step launch (a launch of two attempts: ldlt_mf_twin_kernel<.., true>) and decides there whether that step
  // counts; its scalars and verdict go to the host slot `slot_out` (err_ahead, go).  false if such a launch cannot
  // carry it (then nothing was armed).  ipm_ride_wait(): the verdict, once those scalars are in.
  bool ipm_ride_errors_in_next_step(int slot_out);
  bool ipm_ride_possible();  // (asked once per system: both attempts' tasks and the riding workgroups resident at once; SLPX_IPM_RIDE=0: never)
  bool ipm_ride_wait(int slot_out);
  bool ipm_ride_armed() const { return m_ride_next; }
  void ipm_ride_disarm() { m_ride_next = false; }
  // ipm_errors(.., ahead) that also takes the decision
  void ipm_errors_deciding(bool sums_ride);
  unsigned long long seq_expected() const { return m_seq_expected; }
  void wait_published_until(unsigned long long seq);
  // what a launch changes on the host side of the step machinery (buffer parities, pending work): saved before a
  // speculative launch, put back when the device let it pass
  struct LaunchBook {
    int stats_cur, stats_tw_cur, xg_parity, xg_tw_parity, twin_mode, kkt_pending;
    bool last_step_chained, stats_in_host, lhs_stale, rhs_stale, tape_pending, touched;
    unsigned long long stats_seq;
  };
  LaunchBook save_book() const {
    return LaunchBook{m_stats_cur, m_stats_tw_cur, m_xg_parity, m_xg_tw_parity, m_twin_mode, m_kkt_pending, m_last_step_chained,
                      m_stats_in_host, m_lhs_stale, m_rhs_stale, m_stream.tape_pending, m_stream.touched, m_stats_seq};
  }
  // launch_ran: the launch was not let pass at its gate — it ran with its results held back (a step that carried the
  // error launch deciding about it: MfDev::ride_verdict) and so moved the launch's OWN hand-overs on: the buffers the
  // backward solve hands x through have changed roles for good
  void restore_book(const LaunchBook& b, bool launch_ran = false) {
    m_stats_cur = b.stats_cur;
    m_stats_tw_cur = b.stats_tw_cur;
    if (!launch_ran) {
      m_xg_parity = b.xg_parity;
      m_xg_tw_parity = b.xg_tw_parity;
    }
    m_twin_mode = b.twin_mode;
    m_kkt_pending = b.kkt_pending;
    m_last_step_chained = b.last_step_chained;
    m_stats_in_host = b.stats_in_host;
    m_lhs_stale = b.lhs_stale;
    m_rhs_stale = b.rhs_stale;
    m_stream.tape_pending = b.tape_pending;
    m_stream.touched = b.touched;
    m_stats_seq = b.stats_seq;
  }
  double* d_V_trial() { return m_V_trial.p; }
  // the tape's separable sums (for a caller's launch that lets them ride: restoration.hip's error launch)
  const NlpStructure::SumReduce* reduces_dev() const { return m_reduces.p; }
  int n_reduces() const { return static_cast<int>(m_reduces.n); }
  const double* tape_scales_dev() const { return m_scales.p; }

  // device pointers for callers that keep everything resident
  double* d_x() { return m_in.p; }
  double* d_V() { return m_V.p; }
  double* d_s() { return m_s.p; }
  double* d_y() { return m_y.p; }
  double* d_z() { return m_z.p; }
  double* d_mu() { return m_mu.p; }
  // (a step whose factorization evaluated the system in place left nothing in memory: assembled on demand)
  double* d_lhs() {
    materialize_kkt();
    materialize_batch_major();
    return m_lhs.p;
  }
  double* d_rhs() {
    materialize_kkt();
    materialize_batch_major();
    return m_rhs.p;
  }
  void materialize_kkt();
  void materialize_batch_major();
  double* d_p() { return m_p.p; }
  // A caller that writes the system itself (restoration.hpp: the reduced system of the restoration problem on this
  // system's pattern): the arrays as they are, and "what is there IS the current system" / "the state changed under it"
  const KktDev& kdev() const { return m_kdev; }
  double* lhs_raw() { return m_lhs.p; }
  double* rhs_raw() { return m_rhs.p; }
  void system_written_by_caller(bool lhs, bool rhs) {
    m_kkt_pending = 0;
    if (lhs) m_lhs_stale = m_lhs_in_il = false;
    if (rhs) m_rhs_stale = m_rhs_in_il = false;
  }
  void state_changed_by_caller() { m_lhs_stale = m_rhs_stale = true; }
  double* d_trial_in() { return m_trial_in.p; }
  double* d_s_ahead() { return m_s_ahead.p; }
  double* d_y_ahead() { return m_y_ahead.p; }
  double* d_z_ahead() { return m_z_ahead.p; }
  hipStream_t raw_stream() const { return m_stream.raw(); }
  double* d_ps() { return m_ps.p; }
  double* d_pz() { return m_pz.p; }
  double* d_D() { return m_D.p; }
  double* d_Lx() { return m_Lx.p; }
  int in_stride() const { return m_s_ref.n_inputs(); }
  int v_stride() const { return m_s_ref.nV; }

  const NlpStructure& structure() const { return m_s_ref; }
  const KktPlan& kkt() const { return m_k_ref; }
  const LdltPlan& ldlt() const { return m_l_ref; }

 private:
  void launch_tape(const TapeDevice& t, bool reverse);
  void launch_tape(const TapeDevice& t, bool reverse, hipStream_t small_stream, hipStream_t other);
  void write_reg(const std::vector<double>& delta, const std::vector<double>& gamma,
                 const std::vector<uint8_t>& active);
  void enqueue_factor(int parity, hipStream_t stream);

  DeviceArena m_arena;  // the plan arrays uploaded by the constructor (DevBuf::upload places them here)
  const NlpStructure& m_s_ref;
  const KktPlan& m_k_ref;
  const LdltPlan& m_l_ref;
  int m_batch;
  int m_device;
  TrackedStream m_stream;

  TapeDevice m_full, m_values;
  DevBuf<NlpStructure::SumReduce> m_reduces;
  // KKT plan
  DevBuf<int32_t> m_dptr, m_dsrc, m_pptr, m_pa, m_pb, m_pr, m_gsrc, m_ae_colptr, m_ae_rowidx,
      m_ai_colptr, m_ai_rowidx, m_ai_rowptr, m_ai_col, m_ai_src, m_diag_pos, m_fast_src;
  KktDev m_kdev{};
  // LDLT plan
  DevBuf<LdltTask> m_ltasks;
  DevBuf<int32_t> m_ent_src, m_perm;
  DevBuf<uint8_t> m_ent_flags;
  DevBuf<uint16_t> m_ent_col;
  DevBuf<uint32_t> m_ent_out, m_ent_pair_ptr, m_ent_contrib_ptr, m_contrib_idx, m_ext_dst, m_llvl_ptr,
      m_col_perm, m_col_lvl_ptr, m_fwd_ptr, m_fwd_contrib_ptr, m_scontrib_idx, m_sext_ptr,
      m_sext_dst, m_bwd_ptr;
  DevBuf<LdltPair> m_pairs;
  DevBuf<LdltSn> m_sn_desc;
  DevBuf<int32_t> m_lhs_colptr, m_lhs_rowidx, m_lhs_rowptr, m_lhs_rowent, m_lhs_rowcol;  // refine_solution (uploaded on first use)
  // the dense branch (LdltPlan::dense, ldlt_dense_kernels.h): the factors of every problem as a dim x dim matrix
  bool m_dense = false, m_dense_pivoted = false;
  DevBuf<int32_t> m_dense_trans;  // the transpositions of the pivoted factorization (LdltPlan::dense_pivoted)
  DevBuf<double> m_dense_A;
  DevBuf<int32_t> m_dense_colptr, m_dense_rowidx;
  uint32_t m_dense_lds = 0;
  DevBuf<double> m_rhs0, m_p_acc;
  DevBuf<uint32_t> m_sn_lvl_ptr, m_col_sn, m_lvl_pack, m_col_lvl_pack;
  DevBuf<uint2> m_bwd_range;
  DevBuf<LdltSolveItem> m_fwd_items, m_sext_items, m_bwd_items;
  LdltDev m_ldev{};
  // per-batch values
  DevBuf<double> m_in, m_in_scale, m_scales, m_V, m_s, m_y, m_z, m_mu, m_lhs, m_rhs, m_p, m_ps,
      m_pz, m_D, m_Lx, m_contrib, m_scontrib, m_zv, m_xg, m_scratch;
  DevBuf<LdltStats> m_stats;  // 2 x batch, double-buffered per factorization attempt
  int m_stats_cur = 0;
  DevBuf<double> m_reg_dev;           // device copy of m_h_reg for big batches
  double* m_h_reg = nullptr;          // pinned, read by the kernels: (delta, gamma) per problem;
                                      // delta = NaN: skip the problem
  LdltStats* m_h_stats = nullptr;     // pinned read-back
  bool m_stats_in_host = false;       // the last launch already copied the counters out
  // one problem: the publishing kernel also bumps a sequence number the host spins on
  volatile unsigned long long* m_h_seq = nullptr;  // pinned
  DevBuf<unsigned long long> m_seq_dev;
  unsigned long long m_seq_expected = 0;  // publishing launches enqueued so far
  unsigned long long m_stats_seq = 0;     // the one that carries the current inertia counters
  bool m_fuse_kkt_store = false;
  bool m_fuse_kkt = false, m_fuse_backsub = false;  // the two halves of it (SLPX_FUSE_KKT, SLPX_FUSE_BACKSUB)
  bool m_fuse_launches = false;       // KKT assembly inside the factorization launch, back-substitution inside the solve's
  bool m_defer_kkt = false;           // build_kkt() only notes the request ...
  int m_kkt_pending = 0;              // ... for enqueue_factor (1: lhs + rhs, 2: + the tape's sums)
  bool m_fuse_solve = false;          // factorization and solve of a step in one launch (ldlt_mf_step_kernel)
  // backward solve: x of finished columns handed to the descendants through the values (two
  // buffers, the solves alternate)
  bool m_xg_by_data = false;
  int m_xg_parity = 0;
  DevBuf<double> m_xg2;
  double* xg_now() { return (m_xg_by_data && m_xg_parity) ? m_xg2.p : m_xg.p; }
  double* xg_other() {
    if (!m_xg_by_data) return nullptr;
    return m_xg_parity ? m_xg.p : m_xg2.p;
  }
  void xg_flip() {
    if (m_xg_by_data) m_xg_parity ^= 1;
  }
  DevBuf<unsigned int> m_exit_cnt;
  void enqueue_factor_solve(int parity);
  KktFuse take_kkt_fuse();
  BacksubFuse backsub_fuse(const LdltStats* publish);
  // multifrontal step (ldlt_mf_kernels.h: ldlt_mf_step_kernel; SLPX_LDLT_MF=0: the pair lists)
  bool m_mf = false;
  bool m_mf_solve = false;  // a new right-hand side goes through the fronts too (ldlt_mf_solve_kernel; SLPX_MF_SOLVE=0: the pair lists)
  bool m_mf_mfma = false;             // the plan has fronts on the matrix cores: the kernel variant with that path
  int m_mf_threads = 1024;            // 512 where the 1024-thread workgroups of every task are not resident at once
  int m_twin_threads = 1024;          // ... of a launch of two attempts (twin_available)
  uint32_t m_mf_lds = 0;
  // host copies of the inline KKT / back-substitution plans (build_mf packs them into the task images)
  std::vector<int32_t> m_h_vsrc;
  std::vector<KktTerm> m_h_terms;
  std::vector<uint2> m_h_task_terms;
  std::vector<BsRow> m_h_bs_plan;
  std::vector<uint4> m_h_bs_task_plan;
  DevBuf<LdltMfTask> m_mf_tasks;
  DevBuf<LdltFront> m_mf_fronts;
  uint32_t m_mf_image_stride16 = 0;
  DevBuf<uint4> m_mf_image;        // per task: everything static it keeps in LDS, in LDS order (one copy loop)
  DevBuf<uint4> m_mf_image_desc;   // per task {first 16-byte group, groups up to the end of the KKT terms, groups of back-substitution rows, terms groups}
  DevBuf<double> m_mf_contrib;
  // the second attempt of a twin launch: its own factor, update slots, x hand-over (a pair, alternating like
  // m_xg / m_xg2), direction and counters (a pair, alternating like m_stats)
  KktFuse kkt_fuse_for(int kkt_mode) const;
  void launch_mf_step(int twin_mode, const double* reg, const KktFuse& f, bool chained, const double* lhs2 = nullptr,
                      const double* rhs2 = nullptr);
  void book_mf_step(int twin_mode, bool chained);
  IpmLookaheadArgs lookahead_args(double tau, int twin_mode);
  int m_twin_state = 0;  // 0: not looked at yet, 1: available, -1: not (not resident at once, SLPX_TWIN=0, ...)
  int m_twin_mode = 0;   // of the step launch in flight (IpmTwin::mode; 0: a single attempt)
  DevBuf<double> m_Lx_tw, m_D_tw, m_zv_tw, m_p_tw, m_ps_tw, m_pz_tw, m_mf_contrib_tw, m_xg_tw, m_xg2_tw;
  DevBuf<LdltStats> m_stats_tw;
  int m_stats_tw_cur = 0, m_xg_tw_parity = 0;
  void build_mf(const LdltPlan& l);
  DevBuf<unsigned int> m_ipm_err_done;
  DevBuf<BsRow> m_bs_plan;            // BacksubFuse::plan (rows and terms share the 8-byte element size)
  DevBuf<uint4> m_bs_task_plan;
  uint32_t m_solve_lds_inline = 0;    // dynamic LDS of the backward solve with the rows staged
  void build_inline_backsub(const KktPlan& k, const LdltPlan& l);
  DevBuf<int32_t> m_ent_vsrc;
  DevBuf<KktTerm> m_kkt_terms;
  DevBuf<uint2> m_task_terms;
  uint32_t m_factor_lds_inline = 0;   // dynamic LDS of the factorization with the terms staged
  void build_inline_kkt(const NlpStructure& s, const KktPlan& k, const LdltPlan& l);
  bool m_lhs_stale = false, m_rhs_stale = false;  // memory does not hold the system of the current state
  void solve_after_factor_impl(const LdltStats* publish);
  // all rounds of a factorization / backward solve in one launch (device-side round
  // counters, double-buffered like the inertia counters); SLPX_SINGLE_LAUNCH=0 disables
  bool m_single_launch = true;
  // batch-interleaved LDLT (ldlt_il_kernels.h)
  bool m_il = false, m_il_outputs_stale = false;
  // the assembly kernels write the interleaved lhs / rhs themselves; *_in_il: the current system is in the
  // interleaved arrays only (a caller that wrote its own batch-major system: il_gather_kernel transposes it)
  bool m_lhs_in_il = false, m_rhs_in_il = false;
  DevBuf<double> m_lhs_il, m_rhs_il, m_Lx_il, m_D_il, m_contrib_il, m_scontrib_il, m_zv_il, m_xg_il;
  DevBuf<LdltStats> m_stats_part;  // [task][problem]
  DevBuf<uint32_t> m_il_meta, m_il_meta_off;  // per task: the plan slices the factor kernel stages
  uint32_t m_il_factor_lds = 0, m_il_solve_lds = 0;
  bool m_slot_handoff = false;        // factorization rounds hand over through the update block slots
  DevBuf<uint32_t> m_round_ptr;
  DevBuf<unsigned int> m_fround_cnt, m_bround_cnt;  // [2][batch][n_rounds]
  std::vector<double> m_V_static;  // scaled static values (host copy)
  // interior-point iteration state (ipm_enable)
  bool m_ipm = false;
  DevBuf<double> m_s_ahead, m_y_ahead, m_z_ahead;  // with m_trial_in (x | y | z) and m_V_trial: the look-ahead iterate
  DevBuf<double> m_trial_in, m_V_trial, m_soc_ce, m_soc_cims, m_p_keep, m_ps_keep, m_pz_keep, m_ipm_alpha, m_ipm_scales, m_ipm_partial;
  IpmHost* m_ipm_host = nullptr;   // pinned, two slots (ipm_set_slot)
  int m_ipm_slot = 0;
  struct IpmCtl* m_ipm_ctl_host = nullptr;   // pinned: [0] the device's mirror, [1], [2] staging of uploads
  int m_ipm_ctl_stage = 0;
  void* m_ipm_ctl_dev = nullptr;             // IpmCtl on the device
  DevBuf<double> m_ipm_gate;
  bool m_gate_next_step = false;
  std::vector<double> m_reg_shadow;  // what m_reg_dev holds (enqueue_factor)
  bool m_ride_next = false;
  int m_ride_slot = 0;
  double m_ride_ticket = 1.0;       // of the last riding launch (2, 3, ...: never a plain launch's 0 / 1 in IpmHost::go)
  DevBuf<double> m_ipm_ride_verdict;
  int m_ride_lds_ok = -1;           // the riding kernels' dynamic LDS was raised / fits (-1: not asked yet)
  bool m_errors_decide = false;
  bool m_tape_reduce = true;       // launch_tape runs the separable-sum reductions itself
  // chained steps (sweep_full_for_step): words 0 / 16 / 32 / 48 of m_chain = workgroups of the sweep
  // through, last step whose sweep is complete, workgroups of the step kernel through, last step whose
  // kernel is complete
  bool m_chain_on = false;
  int m_chain_mode = 0;               // TapeJitOptions::chain_mode the full tape's kernel was generated with
  bool m_last_step_chained = false;   // the last multifrontal step kernel launched signals its end through the chain words
  hipStream_t m_tape_stream = nullptr;
  hipEvent_t m_chain_ev = nullptr;
  DevBuf<unsigned int> m_chain;
  int m_chain_failures = 0;
  bool m_debug_break_chain = false;
  unsigned int m_chain_seq = 0;       // number of the last chained step
  // running totals of the workgroups of chained sweeps / chained step kernels launched: what the kernels wait for
  // in chain[16] / chain[48] (kept below 2^29: bits 30, 31 of the words are the failure flag)
  unsigned int m_chain_sweep_wgs = 0, m_chain_step_wgs = 0, m_last_tape_workgroups = 0;
  struct ChainArgs {
    unsigned int* chain = nullptr;
    unsigned int wait_step = 0, this_step = 0;
    bool skip_flag = false;  // `chain` is the look-ahead chain's "rejected attempt" flag, not a chain buffer
  };
  ChainArgs m_chain_args;             // what launch_tape hands the generated kernel (null: an ordinary launch)
  const double* m_in_override = nullptr;  // launch_tape reads / writes these when set
  double* m_V_override = nullptr;
};

}  // namespace slpx
