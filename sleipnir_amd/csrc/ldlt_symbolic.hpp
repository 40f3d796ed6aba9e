// Symbolic phase of the sparse LDLᵀ (host, once per sparsity pattern) — the
// counterpart of Eigen::SimplicialLDLT::analyzePattern as the reference uses it
// (include/sleipnir/optimization/solver/util/sparse_regularized_ldlt.hpp:69-72):
// fill-reducing ordering, elimination tree, pattern of L — plus everything the
// MI355X numeric kernels need so that they contain no searching at all:
//
//   * ordering = nested dissection on the KKT graph (BFS level-structure
//     separators) with minimum-degree leaves.  A chain-structured KKT system
//     (direct transcription) gets an elimination tree of height O(log N) instead
//     of the O(N) chain a band/AMD ordering produces; on a GPU depth, not fill,
//     is the cost (SURVEY.md §7 hard part 2).
//   * the etree is cut into subtrees whose columns fit in LDS ("tasks"); a
//     workgroup factors its subtree level by level entirely in LDS.  Tasks are
//     grouped in rounds (a task only depends on tasks of earlier rounds); one
//     kernel launch per round, no intra-launch cross-workgroup traffic.
//   * numeric factorization is a pure gather: entry (i,j) of column j is
//     A(i,j) − Σ_k U(i,k)·U(j,k)/d_k over an explicit, precomputed list of k
//     (left-looking, entry-parallel).  Updates that cross a task boundary are
//     pre-summed by the child task into a "contribution" slot (multifrontal-style
//     update block) so a parent never reads a child's L.
//   * triangular solves reuse the same tasks/rounds: forward bottom-up with
//     contribution slots, backward top-down reading finished ancestors.
#pragma once

#include <cstdint>
#include <exception>
#include <vector>

#include "nlp.hpp"

namespace slpx {

struct LdltPair {
  uint16_t a, b;  // local entry indices: U(i,k), U(j,k)
  uint16_t k;     // local column index of k (for 1/d_k)
  uint16_t pad;
};

struct LdltTask {
  uint32_t n_ent, n_col, n_ext;
  uint32_t ent_off;    // into per-entry arrays
  uint32_t col_off;    // into per-column arrays
  uint32_t lvl_off;    // into lvl_ptr (entries) / col_lvl_ptr (columns)
  uint32_t n_lvl;
  uint32_t ext_off;    // into ext arrays
  uint32_t pair_off;   // base index into pairs
  uint32_t contrib_off;  // base index into contrib_idx
  // solve
  uint32_t fwd_item_off, bwd_item_off, sext_off, n_sext, sext_item_off, scontrib_off;
  uint32_t round;
  // bases of this task's slices in the *_ptr arrays (each slice ends in a sentinel)
  uint32_t pair_ptr_off;     // ent_pair_ptr: n_ent + n_ext + 1 entries
  uint32_t contrib_ptr_off;  // ent_contrib_ptr: n_ent + 1 entries
  uint32_t colptr_off;       // fwd_ptr / fwd_contrib_ptr / bwd_ptr: n_col + 1 entries
  uint32_t sext_ptr_off;     // sext_ptr: n_sext + 1 entries
  uint32_t n_pairs;          // update pairs of own + pseudo entries
  uint32_t n_fwd_items, n_bwd_items;
  // supernodes of two or more columns (see LdltSn): descriptors in level order
  uint32_t sn_off, n_sn;  // this task's slice of `sn_desc`; `sn_lvl_ptr` shares lvl_off
  uint32_t n_contrib_idx;  // length of this task's slice of contrib_idx (padded to 4 in the array)
};

// A supernode of w >= 2 columns: a chain j_0 < ... < j_{w-1} of the elimination tree inside
// one task whose columns have the SAME structure below the chain (struct(L_{j_c}) =
// {j_{c+1}, ..., j_{w-1}} u R) — the separator cliques of a dissected transcription problem,
// whatever the number of children a column has (relaxed in that sense; no explicit zeros).
// The chain is ONE level of the task: first every entry of its columns receives the updates
// of the columns below the chain (the pair lists, which contain no in-chain pair), then the
// dense (w + |R| + 1) x w trapezoid — diagonal block, rows R, right-hand-side row — is
// finished in registers by ONE WAVE, a lane per row (ldlt_kernels.h: sn_finish_wave).  Its entries are contiguous in the task's local numbering: row t
// of column c (t >= c; t = c the diagonal, t = nr - 1 the rhs row) is entry
// base0 + c nr - c (c - 1) / 2 + (t - c).
struct LdltSn {
  uint32_t base0;  // local entry index of the first column's diagonal
  uint16_t w;      // columns
  uint16_t nr;     // rows: w + |R| + 1
  uint16_t col0;   // local column index of j_0 (the chain's columns are consecutive)
  uint16_t pad;
};
constexpr uint32_t kSnWidthMax = 8;  // longer chains are cut (a lane keeps a row of this many doubles in registers)
constexpr uint32_t kSnRowsMax = 64;   // rows of the trapezoid (w + |R| + 1): one lane each; bigger ones stay single columns

// ---------------------------------------------------------------------------------------------
// Multifrontal plan (ldlt_mf_kernels.h): every supernode of a task — lone columns included — is a
// dense FRONT: rows = [its w columns | the common structure R below them | the right-hand-side
// row], nr = w + r + 1; its first w columns are the LdltSn trapezoid in the task's entry array.
// What the pair lists did entry by entry (left-looking), fronts do block by block: a front's update
// block S (rows R + rhs, columns R; packed, entry (a, b) at a (a + 1) / 2 + b) is handed to its
// parent front inside the task, whose entries take at most `nch` such values each, or — a front
// whose parent column lies in another task — straight to that task's update slots.  Everything a
// lane needs is a precomputed 16-bit BYTE offset into the first 64 KB of the task's LDS:
//   [ U: n_ent doubles | arena: 0.0, a scratch double, the S blocks | 1/d: n_col | x: n_col + n_anc + 1 ]
// Tables of a front in mf_tab (16-bit words), at LdltFront::tab:
//   pivot table   nr rows x (1 + nch) x w : [k][c], k = 0 the entry of U itself (rows above the diagonal:
//                 the scratch double), k >= 1 the k-th child value (none: the 0.0)
//   update table  n_s entries x (3 + nch) : out (the S slot), U(row a, column 0), U(row b, column 0), children
//   solve table   r words                 : where x of row t of R lives (own column, or an ancestor task's row)
// ---------------------------------------------------------------------------------------------
struct LdltFront {
  uint32_t tab;      // first word of its tables in the task's slice of mf_tab
  uint16_t base0;    // local entry of the first column's diagonal
  uint16_t col0;     // local column of the first column
  uint8_t w, nr, nch, flags;  // flags bit 0: its update block leaves the task (mf_ext); bit 1: matrix-core path
  uint16_t n_s;      // entries of its update block: r (r + 1) / 2 + r
  uint16_t ext;      // first slot of its block in the task's slice of mf_ext (flags bit 0)
};
static_assert(sizeof(LdltFront) == 16, "staged into LDS as one 16-byte group");
struct LdltMfTask {
  uint32_t front_off, n_front;   // slice of mf_fronts (level order; mf_lvl_ptr shares LdltTask::lvl_off)
  uint32_t tab_off, n_tab;       // slice of mf_tab, 16-bit words (padded to 8)
  uint32_t ext_off, n_ext;       // slice of mf_ext (padded to 4)
  // update slots from child tasks: only the entries that take any (n_cent of them, mf_cent: local entry
  // indices, padded to 8; mf_contrib_ptr: n_cent + 1, padded to 4), their slots in mf_contrib_idx
  uint32_t cent_off, n_cent;
  uint32_t contrib_ptr_off;
  uint32_t contrib_off, n_contrib_idx;  // slice of mf_contrib_idx (padded to 4)
  uint32_t anc_off, n_anc;       // slice of mf_anc: rows of ancestor tasks its fronts reach (permuted indices)
  uint32_t arena;                // doubles: 0.0, scratch, update blocks
};

struct LdltSolveItem {
  uint32_t lpos;  // index into Lx
  uint32_t ref;   // fwd: local column of y_k; bwd: local column (bit31 clear) or global permuted row (bit31 set)
};

struct LdltPlan {
  int n = 0;        // matrix order
  int n_dec = 0;    // leading block regularized with +δ (decision variables)
  std::vector<int32_t> perm, iperm;  // perm[new] = old
  std::vector<int32_t> parent;       // etree (permuted space)
  std::vector<int32_t> Lp, Li;       // strictly-lower pattern of L, CSC, permuted space
  int64_t nnzL = 0;
  int etree_height = 0;
  int n_rounds = 0;
  bool structurally_singular_unregularized = false;  // some pivot is structurally 0 when δ=γ=0

  std::vector<LdltTask> tasks;                // sorted by round
  std::vector<uint32_t> round_ptr;            // n_rounds + 1, into tasks
  uint32_t max_lds_doubles = 0;               // factor kernel
  uint32_t max_solve_lds_doubles = 0;
  uint32_t factor_lds_bytes = 0;              // dynamic LDS of ldlt_factor_kernel
  uint32_t solve_lds_bytes = 0;               // dynamic LDS of ldlt_fwd/bwd kernels

  // ---- factor ----
  std::vector<int32_t> ent_src;       // index into lhs values or -1
  std::vector<uint8_t> ent_flags;     // bit0: diagonal, bit1: (diagonal) regularize with −γ instead of +δ,
                                      // bit2: right-hand-side row (ent_src indexes rhs, ent_out = permuted column of z)
  std::vector<uint16_t> ent_col;      // local column
  std::vector<uint32_t> ent_out;      // diag: permuted column index (D); else position in Lx
  std::vector<uint32_t> ent_pair_ptr;     // per task: n_ent + n_ext + 1 (relative to pair_off)
  std::vector<uint32_t> ent_contrib_ptr;  // per task: n_ent + 1 (relative to contrib_off)
  std::vector<uint32_t> contrib_idx;      // indices into the contribution buffer
  std::vector<uint32_t> ext_dst;          // contribution-buffer slot of each pseudo entry
  std::vector<uint32_t> lvl_ptr;          // per task n_lvl + 1 (local entry indices)
  std::vector<LdltPair> pairs;
  uint32_t n_contrib = 0;

  // ---- solve ----
  std::vector<uint32_t> col_perm;         // per column (task order): permuted column index
  std::vector<uint32_t> col_lvl_ptr;      // per task n_lvl + 1 (local column indices)
  std::vector<uint32_t> fwd_ptr;          // per task n_col + 1 (relative to fwd_item_off)
  std::vector<uint32_t> fwd_contrib_ptr;  // per task n_col + 1 (relative to scontrib_off)
  std::vector<uint32_t> scontrib_idx;     // indices into the solve contribution buffer
  std::vector<LdltSolveItem> fwd_items;
  std::vector<uint32_t> sext_ptr;         // per task n_sext + 1 (relative to sext_item_off)
  std::vector<uint32_t> sext_dst;
  std::vector<LdltSolveItem> sext_items;
  std::vector<uint32_t> bwd_ptr;          // per task n_col + 1 (relative to bwd_item_off)
  std::vector<LdltSolveItem> bwd_items;
  uint32_t n_scontrib = 0;

  // ---- supernodes (w >= 2 only; none when LdltOptions::supernodal is off) ----
  std::vector<LdltSn> sn_desc;            // per task, level order
  std::vector<uint32_t> sn_lvl_ptr;       // per task n_lvl + 1 (relative to sn_off), in step with lvl_ptr
  // solves: per column (task order) pos-in-chain | w << 8 (a lone column: 0x100)
  std::vector<uint32_t> col_sn;
  int n_supernodes = 0;                   // all of them, singletons included
  int widest_supernode = 1;
  int critical_levels = 0;                // sum over rounds of the deepest task's level count
  std::vector<int32_t> sn_width_hist;     // [w] = supernodes of that width

  // ---- multifrontal plan (LdltFront above); mf == false: not built, or a task does not fit its addressing ----
  bool mf = false;
  std::vector<LdltMfTask> mf_tasks;       // in step with `tasks`
  std::vector<LdltFront> mf_fronts;
  std::vector<uint32_t> mf_lvl_ptr;       // per task n_lvl + 1 first fronts (relative), in step with lvl_ptr
  std::vector<uint16_t> mf_tab;
  std::vector<uint32_t> mf_ext;
  std::vector<uint32_t> mf_contrib_ptr, mf_contrib_idx;
  std::vector<uint16_t> mf_cent;
  std::vector<uint32_t> mf_anc;
  uint32_t mf_n_contrib = 0;
  uint32_t mf_max_nch = 0, mf_max_front_rows = 0, mf_n_mfma = 0;

  // ---- dense plan (build_dense_ldlt_plan): no tasks, no lists — the matrix is factored as a dense one in
  // memory (ldlt_dense_kernels.h), the reference's dense branch (util/dense_regularized_ldlt.hpp:59-136,
  // chosen there by density, interior_point.hpp:340-352; here also whenever a column of L does not fit a task)
  bool dense = false;
  // ... chosen by the reference's own rule (interior_point.hpp:340-352: the lower triangle fills a quarter of the system
  // or more) and factored the way the reference factors it there: Eigen::LDLT's diagonal pivoting
  // (ldlt_dense_pivoted_factor_kernel).  false with `dense`: the plain dense kernel (a system too big for the sparse plan)
  bool dense_pivoted = false;

  // traffic model: factor = 12k + 16ℓ, solve = 32ℓ + 16 n
  int64_t factor_bytes = 0, solve_bytes = 0;
  int64_t flops = 0;  // 2 * number of pair products
  size_t n_pairs() const { return pairs.size(); }
};

struct LdltOptions {
  // Nested-dissection leaves (nodes).  Dissecting down to tiny leaves gives the shallowest
  // trees on chain-structured KKT systems (cart-pole N=1000: height 50 and nnz(L) 67.8 k
  // at 8, height 64 / 72.8 k at 48, height 90 at 96).
  int leaf_size = 8;
  // LDS budget per task in entry-equivalents (one L entry ~ 32 B incl. its
  // descriptors; four update pairs ~ one entry).  2048 keeps a task near 64-96 KB.
  uint32_t task_entries = 2048;
  // One problem, every round in one launch (NewtonSystem): a partition of more than 400 tasks is redone with twice
  // the task size, one of more than 250 keeps the chains-from-the-deepest-child rule out of the leaf tasks (newton.cpp
  // has the measurements).  Decided INSIDE the build, right after the task partition — the ordering, the pattern of L
  // and the relaxed supernodes do not depend on either — where it used to be a second full build (0.18 s at N=5000).
  bool single_problem_task_rules = false;
  bool defer_constraints = true;
  // levels are supernodes (chains of equal-structure columns) instead of single columns:
  // cart-pole N=1000 50 -> 16 levels on the critical path, g-fold N=100 68 -> 16.  The
  // batch-interleaved kernels (ldlt_il_kernels.h) read the column-level plan: off for them.
  bool supernodal = true;
  // shorter chains stay single columns: finishing a chain has a fixed cost of about two column
  // levels (profiles/microbench/chain.hip), so a pair gains nothing (cart-pole N=1000, steps/s:
  // 2 -> 11.39 k, 4 -> 11.52 k, off -> 10.42 k; N=5000: 7.66 k, 7.69 k, 7.08 k)
  int min_supernode_width = 4;
  // chains are cut at this many columns; the device kernels hold a row of kSnWidthMax doubles in
  // registers, wider plans are for the host interpreter's what-if statistics only
  uint32_t max_supernode_width = kSnWidthMax;
  uint32_t max_front_rows = kSnRowsMax;  // rows of a front (w + |R| + 1): the batch kernel with four lanes per problem holds 20
  bool balance_supernode_cuts = false;
  bool chain_from_deepest_child = false;  // a column joins its parent's supernode only if no sibling subtree is as deep as its own
  int chain_from_deepest_min_round = 1;   // ... in tasks of this round and later  // chains longer than that in pieces of equal width (multifrontal plans)
  // Hubs (nodes adjacent to a large part of the graph: a timestep shared by every stage, the
  // dense border of an arrow matrix) are set aside and eliminated last, like the "dense rows" of
  // the sparse orderings: degree > max(hub_floor, hub_factor x median degree).  build_ldlt_plan
  // retries with a sharper rule before it gives up on a column that does not fit a task.
  double hub_factor = 3.0;
  uint32_t hub_floor = 24;
  // also build the multifrontal plan (LdltFront) next to the pair lists
  bool multifrontal = false;
  // Relaxed supernodes: a supernode whose parent column heads another one may join it at the price of
  // explicit zeros in L (its columns take the structure of the other's) — at most this many per
  // merge, only for the child on the deepest path below its parent, within the width and row
  // limits.  A dense front three columns wider costs a fraction of the levels it replaces
  // (profiles/microbench/front.hip); 0 = exact structures only.
  int relax_zeros = 0;
  // a front's update block goes to the matrix cores from this many entries (and four pivot columns) up
  uint32_t mfma_min_entries = 400;  // (measured on g-fold N=100: 128 -> 78.8 us, 300 -> 68.4, 400 -> 67.3, never -> 67.7)
};

// `lower` = lower-triangular CSC pattern with a full diagonal (KktPlan::lhs).
// Rows/cols [0, n_dec) get +δ, the rest −γ (sparse_regularized_ldlt.hpp:217-224).
// `diag_has_source` (optional, by ORIGINAL index): 0 where the diagonal entry of
// the unregularized matrix is structurally zero (used for the constrained ordering
// and for the structural-singularity flag).
LdltPlan build_ldlt_plan(const CscPattern& lower, int n_dec, const LdltOptions& opt = {},
                         const std::vector<int32_t>* user_perm = nullptr,
                         const std::vector<uint8_t>* diag_has_source = nullptr);

// The dense plan of an order-n system: identity permutation, L the full lower triangle, nothing else.
LdltPlan build_dense_ldlt_plan(const CscPattern& lower, int n_dec);
// what build_ldlt_plan throws when a column of L (or a task's working set) does not fit the LDS of a CU
bool ldlt_plan_error_is_too_big(const std::exception& e);

}  // namespace slpx
