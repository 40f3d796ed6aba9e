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
  bool defer_constraints = true;
};

// `lower` = lower-triangular CSC pattern with a full diagonal (KktPlan::lhs).
// Rows/cols [0, n_dec) get +δ, the rest −γ (sparse_regularized_ldlt.hpp:217-224).
// `diag_has_source` (optional, by ORIGINAL index): 0 where the diagonal entry of
// the unregularized matrix is structurally zero (used for the constrained ordering
// and for the structural-singularity flag).
LdltPlan build_ldlt_plan(const CscPattern& lower, int n_dec, const LdltOptions& opt = {},
                         const std::vector<int32_t>* user_perm = nullptr,
                         const std::vector<uint8_t>* diag_has_source = nullptr);

}  // namespace slpx
