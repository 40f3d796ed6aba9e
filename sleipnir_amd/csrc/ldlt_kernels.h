// Task-parallel sparse LDLᵀ on gfx950: numeric factorization + inertia, forward and
// backward substitution.  One workgroup = one task (an elimination-subtree that fits
// in LDS, see ldlt_symbolic.hpp); one launch = one round of independent tasks.
//
// Every kernel first STAGES its task into LDS: the static part (update-pair lists,
// pointers, per-entry descriptors, level bounds) with unrolled 16-byte loads — every
// per-task slice starts on a 16-byte boundary (ldlt_symbolic.cpp) — and the numeric
// part (matrix values, child contributions, the L values a solve will touch) with
// four independent gathers per lane in flight.  After that the dependent chain of a
// level is LDS latency only.  A single direct-transcription KKT system is bound by
// (etree height) × (LDS round trip), not by HBM bandwidth or flops (SURVEY.md §7 hard
// parts 1-3): nnz(L) ≈ 74 k and ≈0.7 MFLOP at N=1000.  Batched, the bound is the HBM
// traffic 12·nnz(lhs) + 16·nnz(L) per factorization and 32·nnz(L) + 16·n per solve
// (SURVEY.md §8d).
//
// Replaces Eigen::SimplicialLDLT::factorize / vectorD / solve as used by
// util/sparse_regularized_ldlt.hpp:74-83,105-109,159-161 and Inertia (inertia.hpp:40-50).
#pragma once

#include <hip/hip_runtime.h>

#include "coherent.h"
#include "device.hpp"
#include "kkt_kernels.h"
#include "tape_kernels.h"  // stage16, q16

namespace slpx {

// Phase clocks (wall_clock64, 100 MHz) of the first task of the selected round in the last
// launch of each kernel: [0,8) factor, [8,16) fwd, [16,24) bwd (debug aid,
// slpx_debug_ldlt_clocks).
__device__ unsigned long long g_ldlt_clocks[24];
// Which task records: LdltDev::clock_task, a kernel ARGUMENT (0xffffffff = nobody, the default).
// It used to be looked up in memory at every clock point (a global holding the round, then
// round_ptr[round]): two dependent trips to memory and a wait for everything outstanding, ten
// times per task — several microseconds of instrumentation on every step's critical path.
#define SLPX_LDLT_CLOCK(k)                                                          \
  if (task_index == L.clock_task && blockIdx.y == 0 && threadIdx.x == 0) \
  g_ldlt_clocks[k] = wall_clock64()

// ---------------------------------------------------------------------------
// Rounds inside ONE launch.  A task of round r needs what the tasks of earlier rounds
// wrote (contribution slots; finished x of ancestors in the backward solve).  Instead of
// a kernel boundary per round, tasks of every round are dispatched together and a task
// waits on a per-problem counter of finished tasks of the previous round.  Tasks are
// sorted by round and workgroups are dispatched in index order, so whatever a waiting
// workgroup depends on is already running or finished — it cannot starve them.  A (never
// expected) time-out marks the factorization bad instead of hanging the GPU.
// Measured at cart-pole N=1000: factorization 80 -> 66 us, backward solve 36 -> 33 us.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void round_wait(const unsigned int* cnt, unsigned int target,
                                           LdltStats* stats_b) {
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) {
        if (stats_b != nullptr) atomicAdd(&stats_b->n_bad, 1 << 20);
        break;
      }
    }
  }
  __syncthreads();
}
// `final_target` != 0 marks the round nobody waits for; its last finisher knows every wait
// of the launch is over and clears the problem's counters for the next launch.
__device__ __forceinline__ void round_signal(unsigned int* cnt_base, int round, int n_rounds,
                                             unsigned int final_target) {
  // Every lane's coherent (write-through) stores must be ACKNOWLEDGED before lane 0 bumps
  // the counter.  A workgroup-scope release fence does not guarantee that here (waves of a
  // workgroup share the CU's L1, so the compiler need not drain vmcnt for it) and an
  // agent-scope one adds an L2 write-back; the explicit wait is exactly what is needed.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(&cnt_base[round], 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
    if (final_target != 0 && old + 1 == final_target)
      for (int r = 0; r < n_rounds; ++r)
        __hip_atomic_store(&cnt_base[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Hand-over THROUGH THE DATA (factorization, single launch): an update block slot holds a
// reserved NaN pattern until its producer (one child task) stores the value; its only
// consumer (one entry of the parent task) spins on the slot itself, takes the value and
// re-arms the slot for the next factorization.  Compared with the round counters this
// drops, per round, the producer's wait for its store acknowledgements, a workgroup
// barrier and an atomic increment, and the consumer's separate trip for the data after the
// counter moved.  A computed NaN never carries this payload (the hardware produces the
// canonical quiet NaN), so a numerical breakdown cannot be mistaken for "not yet written".
constexpr unsigned long long kSlotEmpty = 0x7ff8dead0badbeefull;
__device__ __forceinline__ bool slot_is_empty(double v) {
  return static_cast<unsigned long long>(__double_as_longlong(v)) == kSlotEmpty;
}
__device__ __forceinline__ double slot_take(double* p, LdltStats* stats_b) {
  unsigned int spins = 0;
  for (;;) {
    const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (static_cast<unsigned long long>(__double_as_longlong(v)) != kSlotEmpty) {
      __hip_atomic_store(p, __longlong_as_double(static_cast<long long>(kSlotEmpty)), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      return v;
    }
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) {  // never expected: mark the factorization bad instead of hanging
      if (stats_b != nullptr) atomicAdd(&stats_b->n_bad, 1 << 20);
      return 0.0;
    }
  }
}

// The backward solve's version of the same idea: x of a finished column has MANY readers (every
// descendant task that has an entry in that row), so a reader cannot re-arm the slot.  Instead the
// solves alternate between two x buffers, and a task re-arms its own columns in the OTHER buffer
// — the one the next solve will hand over through — after it has published them in this one.
__device__ __forceinline__ double slot_read(const double* p) {
  unsigned int spins = 0;
  for (;;) {
    const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (static_cast<unsigned long long>(__double_as_longlong(v)) != kSlotEmpty) return v;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) return v;  // never expected: the reserved NaN poisons the step instead of hanging
  }
}

// 1/d on the critical path of every level: hardware estimate + two Newton steps (full
// double precision for normal inputs; 0 -> inf and inf -> 0 like the division) instead of
// the ~15-instruction IEEE division sequence with its scale/fixup steps.
__device__ __forceinline__ double fast_reciprocal(double d) {
  const double r0 = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r0, 1.0);
  double r = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return (r0 != 0.0 && isfinite(r0)) ? r : r0;  // d = 0, inf, nan: the estimate is the answer
}

// Sum over the 8 lanes of an aligned lane group with DPP register moves (no LDS crossbar
// trip, unlike __shfl_xor's ds_bpermute): mirror
// within the 8-lane half row, then the two quad permutes.  Lane 0 of the group (and in
// fact every lane) ends up with the group total.
template <int kCtrl>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group8_sum(double v) {
  v += dpp_move<0x141>(v);  // row_half_mirror: lane i <- lane 7 - i
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  return v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/d for the pivots INSIDE a supernode: the same estimate + two Newton steps without the
// special-case select (three more dependent operations on the chain): d = 0 gives NaN instead
// of inf — either way every later pivot is non-finite, and the zero pivot itself is counted
// as bad (n_bad), which is all the policy loop looks at.
__device__ __forceinline__ double chain_reciprocal(double d) {
  const double r0 = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r0, 1.0);
  double r = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}

// ---------------------------------------------------------------------------
// Supernodes (LdltSn, ldlt_symbolic.hpp).  When the level loop reaches a chain of w >= 2
// columns, every entry of its (w + |R| + 1) x w trapezoid already holds A - (updates of the
// columns below the chain).  What is left is dense: U(t,j) -= sum_{c<j} U(t,c) U(j,c) / d_c.
// ONE WAVE PER CHAIN, a lane per row with its row in registers; the pivot row's entries reach
// the other lanes as scalars (v_readlane), so the whole elimination is register traffic —
// w(w-1)/2 broadcast + FMA steps, no LDS round trip and no barrier between two columns.
// Lanes [0, w) hold the block's rows, lanes [w, nr) the rows below it and the
// right-hand-side row (nr <= 64).  The wave is the only reader of the chain's entries in this
// pass, so it stores everything back to U itself; nothing goes to global memory here (a global
// store inside the level loop costs its acknowledgement at the next barrier: measured 82 vs
// 59 us per factorization).
//
// What this buys, measured (profiles/microbench/{latency,chain}.hip on the MI355X): a lone wave
// issues one instruction of such a dependent sequence every ~9 clocks, whatever the
// instruction (a dependent v_fma_f64 alone is 4-6 clocks, a v_readlane pair + FMA 25, an LDS
// round trip 60, a barrier of sixteen waves 64): a column level of the generic gather costs
// ~1140 clocks because it EXECUTES ~75 instructions per wave, not because it waits.  A chain
// finished here costs 1480 (w = 4), 2600 (w = 8) clocks warm and ~1100 more the first time a
// CU runs that width (instruction-cache misses after the jump), against w x 1140 as w levels.
// Variants tried and dropped, clocks per level at N=1000 in the kernel: every row lane factors
// the block itself, no communication (~2800: 150 instructions per lane); one 16-wide body with
// `c < w` guards (5700: every guard is executed); a rolled loop with the row registers indexed
// through M0 (4000-10000); a rolled loop with shifted registers and bodies of 4 / 8 / 16 steps
// (300-780 per COLUMN: each column runs all steps of its body).  Kept: exact-width
// straight-line code per w, reached by one scalar jump, out of line so that its registers do
// not weigh on the level loop.
// ---------------------------------------------------------------------------
// The chain code works on LDS through pointers that SAY so (address space 3): handed a plain
// `double*` an out-of-line function gets a generic pointer and every access becomes a flat
// load / store (slower, and counted on both wait counters); and it takes its wave-uniform
// arguments through readfirstlane ITSELF, because the calling convention passes them in vector
// registers — the width switch was a cascade of exec-mask branches, the address arithmetic
// vector work.  Loads are unconditional (a clamped index, then a select): a guarded load is a
// compare, an exec save, a branch and a restore around it.
using LdsF64 = __attribute__((address_space(3))) double;
using LdsU32 = __attribute__((address_space(3))) uint32_t;
using LdsU2 = __attribute__((address_space(3))) uint2;
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* lds_cast(T* p) {
  return (__attribute__((address_space(3))) T*)p;
}
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) const T* lds_cast(const T* p) {
  return (__attribute__((address_space(3))) const T*)p;
}

template <int W>
__device__ __forceinline__ void sn_finish_wave_exact(LdsF64* __restrict__ U, LdsF64* __restrict__ invd, uint32_t base0,
                                                     uint32_t nr, uint32_t col0, uint32_t lane) {
  const uint32_t t = lane < nr ? lane : nr - 1u;  // (idle lanes shadow the last row; they store nothing)
  const bool live = lane < nr;
  double a[W];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    // column c holds rows c .. nr-1; rows above the diagonal (t < c) read the diagonal and drop it
    const uint32_t offc = base0 + c * nr - (c * (c - 1)) / 2;
    const uint32_t row = t >= static_cast<uint32_t>(c) ? t - c : 0u;
    const double v = U[offc + row];
    a[c] = t >= static_cast<uint32_t>(c) ? v : 0.0;
  }
#pragma unroll
  for (int c = 0; c < W; ++c) {
    const double inv = chain_reciprocal(readlane_f64(a[c], c));
    const double lc = a[c] * inv;
    if (lane == static_cast<uint32_t>(c)) invd[col0 + c] = inv;
#pragma unroll
    for (int j = c + 1; j < W; ++j) a[j] = __builtin_fma(-lc, readlane_f64(a[c], j), a[j]);
  }
#pragma unroll
  for (int c = 0; c < W; ++c)
    if (live && static_cast<uint32_t>(c) <= lane) U[base0 + c * nr - (c * (c - 1)) / 2 + (lane - c)] = a[c];
}

// (every argument but `lane` is wave-uniform)
__device__ __attribute__((noinline)) void sn_finish_wave(LdsF64* __restrict__ U, LdsF64* __restrict__ invd,
                                                         uint32_t base0, uint32_t w, uint32_t nr, uint32_t col0,
                                                         uint32_t lane) {
  base0 = __builtin_amdgcn_readfirstlane(base0);
  w = __builtin_amdgcn_readfirstlane(w);
  nr = __builtin_amdgcn_readfirstlane(nr);
  col0 = __builtin_amdgcn_readfirstlane(col0);
  switch (w) {
#define SLPX_SN_CASE(W) case W: sn_finish_wave_exact<W>(U, invd, base0, nr, col0, lane); break;
    SLPX_SN_CASE(2) SLPX_SN_CASE(3) SLPX_SN_CASE(4) SLPX_SN_CASE(5) SLPX_SN_CASE(6) SLPX_SN_CASE(7) SLPX_SN_CASE(8)
#undef SLPX_SN_CASE
    default: break;
  }
  static_assert(kSnWidthMax == 8, "one case per supernode width");
}

// One wave finishes y (forward) or x (backward) of a chain of w <= kSnWidthMax columns: lane c
// owns column c; its couplings to the other columns of the chain sit in registers, indexed by
// the column they multiply, and the finished components are broadcast with v_readlane.
// FORWARD: `ptr` is the row pointer array (uint32_t, n_col + 1): row c's last c items are
// L(j_c, j_0..j_{c-1}).  Backward: `ptr` is the column range table (uint2): the W - c - 1 items
// before .x are L(j_{c+1}..j_{W-1}, j_c).
template <int W, bool FORWARD, typename Ptr>
__device__ __forceinline__ void chain_solve_wave_w(LdsF64* __restrict__ v, const LdsF64* __restrict__ vals,
                                                   const Ptr* __restrict__ ptr, uint32_t i0, uint32_t lane) {
  const bool mine = lane < static_cast<uint32_t>(W);
  const uint32_t c = mine ? lane : 0u;
  uint32_t first;  // position of the coupling to chain column 0 (forward) / c + 1 (backward)
  if constexpr (FORWARD) first = ptr[i0 + c + 1] - c;
  else first = ptr[i0 + c].x - (static_cast<uint32_t>(W) - c - 1u);
  double cf[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const uint32_t kk = static_cast<uint32_t>(k);
    const bool use = FORWARD ? kk < c : kk > c;
    // (clamped: lanes without such a coupling read their first one and drop it)
    const uint32_t at = use ? (FORWARD ? first + kk : first + (kk - c - 1u)) : first;
    const double x = vals[at];
    cf[k] = (use && mine) ? x : 0.0;
  }
  double p = v[i0 + c];
  p = mine ? p : 0.0;
  if (FORWARD) {
#pragma unroll
    for (int k = 0; k < W - 1; ++k) p = __builtin_fma(-cf[k], readlane_f64(p, k), p);
  } else {
#pragma unroll
    for (int k = W - 1; k >= 1; --k) p = __builtin_fma(-cf[k], readlane_f64(p, k), p);
  }
  if (mine) v[i0 + c] = p;
}
template <bool FORWARD, typename Ptr>
__device__ __attribute__((noinline)) void chain_solve_wave(LdsF64* __restrict__ v, const LdsF64* __restrict__ vals,
                                                           const Ptr* __restrict__ ptr, uint32_t i0, uint32_t w,
                                                           uint32_t lane) {
  i0 = __builtin_amdgcn_readfirstlane(i0);
  w = __builtin_amdgcn_readfirstlane(w);
  switch (w) {
#define SLPX_SN_CASE(W) case W: chain_solve_wave_w<W, FORWARD>(v, vals, ptr, i0, lane); break;
    SLPX_SN_CASE(2) SLPX_SN_CASE(3) SLPX_SN_CASE(4) SLPX_SN_CASE(5) SLPX_SN_CASE(6) SLPX_SN_CASE(7) SLPX_SN_CASE(8)
#undef SLPX_SN_CASE
    default: break;
  }
  static_assert(kSnWidthMax == 8, "one case per supernode width");
}

// ---------------------------------------------------------------------------
// Factorization.  Entry (i,j) of column j:  U(i,j) = A(i,j) [+δ | −γ on the
// diagonal] − Σ contributions of child tasks − Σ_k U(i,k)·U(j,k)/d_k, the last sum
// over an explicit pair list (left-looking, entry-parallel).  d_j = U(j,j),
// L(i,j) = U(i,j)/d_j.  Eight lanes cooperate on one entry's pair list.
//
// LDS: pairs[np] 8 B | pptr[n_ent+n_ext+1] u32 | lvl[n_lvl+1] u32 | src[n_ent] i32 |
//      col[n_ent] u16 | flags[n_ent] u8 | out[n_ent] u32 | cptr[n_ent+1] u32 |
//      sn[n_sn] 12 B | (lvl holds entry index | first chain of the level << 16)
//      U[n_ent] f64 | invd[n_col] f64 | counters 32 B      (16-byte groups first)
// A level is a set of SUPERNODES: pass A = the gather above over every entry of the level's
// columns (in-chain pairs are not in the lists), then — only in levels that hold a chain of two
// or more columns — a barrier and sn_finish_wave, one wave per chain.
// ---------------------------------------------------------------------------
// What the factorization leaves in LDS for its exit phase.
struct FactorKeep {
  double* U;
  double* invd;
  const uint16_t* col;
  const uint8_t* flags;
  const uint32_t* out;
  int* s_cnt;
  unsigned long long* s_minp;
};

// results + inertia (inertia.hpp:40-50: |d| <= eps counts as zero)
template <int THREADS>
__device__ __forceinline__ void ldlt_factor_exit(const FactorKeep& K, const LdltTask& t, int b,
                                                 double* __restrict__ Lx, double* __restrict__ D,
                                                 double* __restrict__ zv, LdltStats* __restrict__ stats) {
  const int tid = threadIdx.x;
  for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
    const double u = K.U[i];
    if (K.flags[i] & 1) {
      D[K.out[i]] = u;
      const double eps = 2.220446049250313e-16;
      if (u > eps) atomicAdd(&K.s_cnt[0], 1);
      else if (u < -eps) atomicAdd(&K.s_cnt[1], 1);
      else atomicAdd(&K.s_cnt[2], 1);
      if (u == 0.0 || !isfinite(u)) atomicAdd(&K.s_cnt[3], 1);
      else atomicMin(K.s_minp, static_cast<unsigned long long>(__double_as_longlong(fabs(u))));
    } else if (K.flags[i] & 4) {
      zv[K.out[i]] = u * K.invd[K.col[i]];  // z = D⁻¹L⁻¹Pb: the forward solve came for free
    } else {
      Lx[K.out[i]] = u * K.invd[K.col[i]];
    }
  }
  __syncthreads();
  if (tid < 4 && K.s_cnt[tid] != 0) atomicAdd(reinterpret_cast<int*>(&stats[b]) + tid, K.s_cnt[tid]);
  if (tid == 0) atomicMin(&stats[b].min_abs_bits, *K.s_minp);
}

// a separable sum of the tape riding in the factorization's launch (they only feed f, which
// nothing in that launch reads)
__device__ __forceinline__ void ride_along_sum(const KktFuse& F, uint32_t which, unsigned char* smem_raw) {
  const int tid = threadIdx.x;
  double* part = reinterpret_cast<double*>(smem_raw);
  const NlpStructure::SumReduce r = F.red[which];
  double acc = 0.0;
  if (tid < 64) {
    for (int k = tid; k < r.count; k += 64) acc += F.V[r.src_off + k];
    part[tid] = acc;
  }
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (tid < w) part[tid] += part[tid + w];
    __syncthreads();
  }
  if (tid == 0) F.Vw[r.dst] = (r.scale_idx >= 0 ? F.scales[r.scale_idx] : 1.0) * part[0];
}

// One task of the factorization.  Returns false when the problem is not part of this attempt.
template <int THREADS>
__device__ __forceinline__ bool ldlt_factor_body(
    const LdltDev& L, uint32_t task_index, bool first_task, const LdltTask& t, int b, unsigned char* smem_raw,
    const double* __restrict__ lhs, int lhs_stride,
    const double* __restrict__ reg, double* __restrict__ Lx, long long lx_stride,
    double* __restrict__ D, int n, double* __restrict__ contrib, int contrib_stride,
    LdltStats* __restrict__ stats, LdltStats* __restrict__ stats_next,
    const double* __restrict__ rhs, double* __restrict__ zv, unsigned int* __restrict__ round_cnt,
    int slot_handoff, const KktFuse& F) {
  const int tid = threadIdx.x;
  // the counters the NEXT factorization attempt accumulates into (nobody touches them now)
  if (stats_next != nullptr && first_task && tid == 0)
    stats_next[b] = LdltStats{0, 0, 0, 0, 0x7ff0000000000000ull};
  const double delta = reg[2 * b], gamma = reg[2 * b + 1];
  if (delta != delta) return false;  // NaN: this problem is not part of the attempt
  lhs += static_cast<size_t>(b) * lhs_stride;
  Lx += static_cast<size_t>(b) * lx_stride;
  D += static_cast<size_t>(b) * n;
  contrib += static_cast<size_t>(b) * contrib_stride;
  rhs += static_cast<size_t>(b) * n;
  zv += static_cast<size_t>(b) * n;

  SLPX_LDLT_CLOCK(0);
  const uint32_t n_pp = t.n_ent + t.n_ext + 1;
  const uint32_t np = t.n_pairs;
  const uint32_t g_pairs = q16(np, 2), g_pptr = q16(n_pp, 4), g_lvl = q16(t.n_lvl + 1, 4),
                 g_src = q16(t.n_ent, 4), g_col = q16(t.n_ent, 8), g_flags = q16(t.n_ent, 16),
                 g_out = q16(t.n_ent, 4), g_cptr = q16(t.n_ent + 1, 4), g_snd = q16(3 * t.n_sn, 4),
                 g_cidx = q16(t.n_contrib_idx, 4);
  uint4* s_pairs = reinterpret_cast<uint4*>(smem_raw);
  uint4* s_pptr = s_pairs + g_pairs;
  uint4* s_lvl = s_pptr + g_pptr;
  uint4* s_src = s_lvl + g_lvl;
  uint4* s_col = s_src + g_src;
  uint4* s_flags = s_col + g_col;
  uint4* s_out = s_flags + g_flags;
  uint4* s_cptr = s_out + g_out;
  uint4* s_snd = s_cptr + g_cptr;
  uint4* s_cidx = s_snd + g_snd;
  double* U = reinterpret_cast<double*>(s_cidx + g_cidx);
  double* invd = U + t.n_ent;
  int* s_cnt = reinterpret_cast<int*>(invd + t.n_col);
  unsigned long long* s_minp = reinterpret_cast<unsigned long long*>(s_cnt + 4);
  const uint2* pairs = reinterpret_cast<const uint2*>(s_pairs);
  const uint32_t* pptr = reinterpret_cast<const uint32_t*>(s_pptr);
  const uint32_t* lvl = reinterpret_cast<const uint32_t*>(s_lvl);
  const int32_t* src = reinterpret_cast<const int32_t*>(s_src);
  const uint16_t* col = reinterpret_cast<const uint16_t*>(s_col);
  const uint8_t* flags = reinterpret_cast<const uint8_t*>(s_flags);
  const uint32_t* out = reinterpret_cast<const uint32_t*>(s_out);
  const uint32_t* cptr = reinterpret_cast<const uint32_t*>(s_cptr);
  const LdltSn* snd = reinterpret_cast<const LdltSn*>(s_snd);

  // ---- stage the static part ----
  stage16<THREADS>(s_pairs, reinterpret_cast<const uint4*>(L.pairs + t.pair_off), g_pairs, tid);
  stage16<THREADS>(s_pptr, reinterpret_cast<const uint4*>(L.ent_pair_ptr + t.pair_ptr_off), g_pptr, tid);
  stage16<THREADS>(s_lvl, reinterpret_cast<const uint4*>(L.lvl_pack + t.lvl_off), g_lvl, tid);
  // (inline assembly, device.hpp KktFuse: what every entry is made of instead of where it sits in lhs)
  stage16<THREADS>(s_src, reinterpret_cast<const uint4*>((F.inline_kkt ? F.ent_vsrc : L.ent_src) + t.ent_off), g_src,
                   tid);
  uint4* s_terms = reinterpret_cast<uint4*>(
      smem_raw + ((static_cast<uint32_t>(reinterpret_cast<unsigned char*>(s_minp + 1) - smem_raw) + 15u) & ~15u));
  uint32_t n_terms16 = 0;  // the task's term block in 16-byte units
  if (F.inline_kkt) {
    const uint2 tt = F.task_terms[task_index];
    n_terms16 = tt.y;
    stage16<THREADS>(s_terms, F.terms + tt.x, tt.y, tid);
  }
  stage16<THREADS>(s_col, reinterpret_cast<const uint4*>(L.ent_col + t.ent_off), g_col, tid);
  stage16<THREADS>(s_flags, reinterpret_cast<const uint4*>(L.ent_flags + t.ent_off), g_flags, tid);
  stage16<THREADS>(s_out, reinterpret_cast<const uint4*>(L.ent_out + t.ent_off), g_out, tid);
  stage16<THREADS>(s_cptr, reinterpret_cast<const uint4*>(L.ent_contrib_ptr + t.contrib_ptr_off), g_cptr,
               tid);
  if (t.n_sn) stage16<THREADS>(s_snd, reinterpret_cast<const uint4*>(L.sn_desc + t.sn_off), g_snd, tid);
  stage16<THREADS>(s_cidx, reinterpret_cast<const uint4*>(L.contrib_idx + t.contrib_off), g_cidx, tid);
  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 0) *s_minp = 0x7ff0000000000000ull;  // +inf
  __syncthreads();
  SLPX_LDLT_CLOCK(1);

  // ---- matrix values: four independent gathers per lane in flight ----
  {
    // entries of the right-hand-side row (flag bit 2) read rhs instead of lhs
    if (!F.inline_kkt) {
      auto fetch = [&](uint32_t i) {
        const int32_t s0 = src[i];
        const double* base = (flags[i] & 4) ? rhs : lhs;
        return s0 >= 0 ? base[s0] : 0.0;
      };
      uint32_t i = tid;
      for (; i + 3 * THREADS < t.n_ent; i += 4 * THREADS) {
        const double a0 = fetch(i), a1 = fetch(i + THREADS), a2 = fetch(i + 2 * THREADS), a3 = fetch(i + 3 * THREADS);
        U[i] = a0;
        U[i + THREADS] = a1;
        U[i + 2 * THREADS] = a2;
        U[i + 3 * THREADS] = a3;
      }
      for (; i < t.n_ent; i += THREADS) U[i] = fetch(i);
    } else {
      // The system is not in memory: every task evaluates the entries it owns from what the AD
      // sweep left in V (each lhs / rhs entry feeds exactly one entry of L, so nothing is
      // computed twice).  What an entry is made of came with the plan, so this is still ONE
      // trip to memory (two for the rare sum of more than kKktUnroll terms) — instead of a whole
      // assembly pass and a kernel boundary in front of this launch.
      // (src[] holds device.hpp's KktFuse::ent_vsrc here)
      const double m = F.mu[0];
      const KktTerm* terms = reinterpret_cast<const KktTerm*>(s_terms);
      double* tprod = reinterpret_cast<double*>(s_terms + n_terms16);
      const uint32_t n_terms = n_terms16 * 4u / 3u;  // (the padding terms are harmless: V[0])
      const uint32_t span = n_terms > t.n_ent ? n_terms : t.n_ent;
      for (uint32_t k = tid; k < span; k += THREADS) {
        // a plain copy and a term per lane and pass, their loads in flight together
        const int w = k < t.n_ent ? src[k] : -1;
        const double v = w >= 0 ? F.V[w & 0x3fffffff] : 0.0;
        const bool on = k < n_terms;
        const KktTermLoads tl = kkt_term_fetch(on ? terms[k] : KktTerm{0, 0, 0}, on, F.V, F.s, F.y, F.z);
        if (k < t.n_ent && w >= -1) U[k] = (w & 0x40000000) && w >= 0 ? -v : v;
        if (on) tprod[k] = kkt_term_product(tl, m);
      }
      __syncthreads();
      for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
        const int w = src[i];
        if (w < -1) {
          const uint32_t code = static_cast<uint32_t>(-(w + 2));
          U[i] = kkt_terms_sum(terms, tprod, code & 0xfffffu, code >> 20, (flags[i] & 4) != 0);
        }
      }
      if (F.store_lhs != nullptr) {  // verification: leave what was evaluated where an assembly pass would have
        __syncthreads();
        for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
          const int32_t s0 = L.ent_src[t.ent_off + i];
          if (s0 >= 0) ((flags[i] & 4) ? F.store_rhs : F.store_lhs)[s0] = U[i];
        }
      }
    }
  }
  __syncthreads();
  // everything above only needed static data and the assembled matrix; the update blocks
  // come from the tasks of earlier rounds
  if (round_cnt != nullptr && !slot_handoff && t.round > 0)
    round_wait(&round_cnt[b * L.n_rounds + t.round - 1],
               L.round_ptr[t.round] - L.round_ptr[t.round - 1], &stats[b]);
  // regularization + update blocks of child tasks (few entries have any)
  {
    // (the slot indices came with the plan: the trip to the slot is the only one on this path)
    const uint32_t* cidx = reinterpret_cast<const uint32_t*>(s_cidx);
    for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
      const uint8_t fl = flags[i];
      const uint32_t cb = cptr[i], ce = cptr[i + 1];
      if (!(fl & 1) && cb == ce) continue;
      double acc = U[i];
      if (fl & 1) acc += (fl & 2) ? -gamma : delta;
      if (slot_handoff) {
        // Four slots in flight per look, ONE loop for every lane (slots past the entry's last are
        // predicated off): an entry of a task with many children would otherwise pay a trip to
        // memory per child even when all of them have long delivered.  Summed in list order.
        for (uint32_t c = cb; c < ce; c += 4) {
          double* p0 = &contrib[cidx[c]];
          double* p1 = &contrib[cidx[c + 1 < ce ? c + 1 : c]];
          double* p2 = &contrib[cidx[c + 2 < ce ? c + 2 : c]];
          double* p3 = &contrib[cidx[c + 3 < ce ? c + 3 : c]];
          double v0, v1, v2, v3;
          unsigned int spins = 0;
          for (;;) {
            v0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v3 = __hip_atomic_load(p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool all_in = !slot_is_empty(v0) && !slot_is_empty(v1) && !slot_is_empty(v2) && !slot_is_empty(v3);
            if (all_in) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {  // never expected: mark the factorization bad instead of hanging
              atomicAdd(&stats[b].n_bad, 1 << 20);
              break;
            }
          }
          const double armed = __longlong_as_double(static_cast<long long>(kSlotEmpty));
          __hip_atomic_store(p0, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc -= v0;
          if (c + 1 < ce) {
            __hip_atomic_store(p1, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc -= v1;
          }
          if (c + 2 < ce) {
            __hip_atomic_store(p2, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc -= v2;
          }
          if (c + 3 < ce) {
            __hip_atomic_store(p3, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc -= v3;
          }
        }
      } else {
        for (uint32_t c = cb; c < ce; ++c) acc -= coherent_load(&contrib[cidx[c]], round_cnt != nullptr);
      }
      U[i] = acc;
    }
  }
  __syncthreads();
  SLPX_LDLT_CLOCK(2);

  // ---- level loop: all in LDS ----
  const int lane8 = tid & 7, grp = tid >> 3;
  {
    // lvl[l] = first entry of level l | first chain (w >= 2) of level l << 16
    uint32_t beg = lvl[0] & 0xffffu, end_pack = t.n_lvl ? lvl[1] : 0, end = end_pack & 0xffffu, sb = 0;
#ifdef SLPX_LDLT_LEVEL_CLOCKS  // core clocks in pass A / in the chain pass (slots 6, 7); each reading costs ~150 clocks
    long long clk_a = 0, clk_b = 0, clk_t = clock64();
#endif
    for (uint32_t l = 0; l < t.n_lvl; ++l) {
      const uint32_t next_pack = lvl[l + 2 <= t.n_lvl ? l + 2 : t.n_lvl];  // one level ahead
      for (uint32_t i = beg + grp; i < end; i += THREADS / 8) {
        const uint32_t pb = pptr[i], pe = pptr[i + 1];
        // issued with the pointer loads, off the dependent chain
        const double u_old = U[i];
        const uint32_t fl = flags[i], cj = col[i];
        double partial = 0.0;
        for (uint32_t q = pb + lane8; q < pe; q += 8) {
          const uint2 pr = pairs[q];
          partial += (U[pr.x & 0xffffu] * invd[pr.y & 0xffffu]) * U[pr.x >> 16];
        }
        partial = group8_sum(partial);
        if (lane8 == 0) {
          const double u = u_old - partial;
          U[i] = u;
          if ((fl & 9) == 1) invd[cj] = fast_reciprocal(u);  // a lone column's pivot
        }
      }
      __syncthreads();
#ifdef SLPX_LDLT_LEVEL_CLOCKS
      {
        const long long c = clock64();
        clk_a += c - clk_t;
        clk_t = c;
      }
#endif
      {
        const uint32_t se = end_pack >> 16;
        if (se > sb) {  // chains of two or more columns in this level (block-uniform)
          // everything about the chain is wave-uniform: scalar registers, scalar jump
          for (uint32_t q = sb + __builtin_amdgcn_readfirstlane(tid >> 6); q < se; q += THREADS / 64) {
            const LdltSn sn = snd[q];
            sn_finish_wave(lds_cast(U), lds_cast(invd), sn.base0, sn.w, sn.nr, sn.col0, tid & 63);
          }
          __syncthreads();
        }
        sb = se;
      }
#ifdef SLPX_LDLT_LEVEL_CLOCKS
      {
        const long long c = clock64();
        clk_b += c - clk_t;
        clk_t = c;
      }
#endif
      beg = end;
      end_pack = next_pack;
      end = end_pack & 0xffffu;
    }
#ifdef SLPX_LDLT_LEVEL_CLOCKS
    if (task_index == L.clock_task && blockIdx.y == 0 && threadIdx.x == 0) {
      g_ldlt_clocks[6] = static_cast<unsigned long long>(clk_a);
      g_ldlt_clocks[7] = static_cast<unsigned long long>(clk_b);
    }
#endif
  }

  SLPX_LDLT_CLOCK(3);
  // ---- update blocks for ancestor tasks (later rounds) ----
  for (uint32_t x = grp; x < t.n_ext; x += THREADS / 8) {
    const uint32_t pb = pptr[t.n_ent + x], pe = pptr[t.n_ent + x + 1];
    double partial = 0.0;
    for (uint32_t q = pb + lane8; q < pe; q += 8) {
      const uint2 pr = pairs[q];
      partial += (U[pr.x & 0xffffu] * invd[pr.y & 0xffffu]) * U[pr.x >> 16];
    }
    partial = group8_sum(partial);
    if (lane8 == 0) coherent_store(&contrib[L.ext_dst[t.ext_off + x]], partial, round_cnt != nullptr);
  }
  // the next round only waits for the update blocks, not for L and D going out
  if (round_cnt != nullptr && !slot_handoff) {
    const int last = L.n_rounds - 1;
    round_signal(&round_cnt[b * L.n_rounds], static_cast<int>(t.round), L.n_rounds,
                 static_cast<int>(t.round) == last ? L.round_ptr[last + 1] - L.round_ptr[last] : 0u);
  }

  SLPX_LDLT_CLOCK(4);
  ldlt_factor_exit<THREADS>(FactorKeep{U, invd, col, flags, out, s_cnt, s_minp}, t, b, Lx, D, zv, stats);
  SLPX_LDLT_CLOCK(5);
  return true;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void ldlt_factor_kernel(
    LdltDev L, uint32_t task_base, const double* __restrict__ lhs, int lhs_stride,
    const double* __restrict__ reg, double* __restrict__ Lx, long long lx_stride,
    double* __restrict__ D, int n, double* __restrict__ contrib, int contrib_stride,
    LdltStats* __restrict__ stats, LdltStats* __restrict__ stats_next,
    const double* __restrict__ rhs, double* __restrict__ zv, unsigned int* __restrict__ round_cnt,
    int slot_handoff, KktFuse F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (static_cast<int>(blockIdx.x) < F.n_blocks) {
    ride_along_sum(F, blockIdx.x, smem_raw);
    return;
  }
  const uint32_t task_index = task_base + blockIdx.x - static_cast<uint32_t>(F.n_blocks);
  const LdltTask t = L.tasks[task_index];
  ldlt_factor_body<THREADS>(L, task_index, task_index == task_base, t, blockIdx.y, smem_raw, lhs, lhs_stride, reg, Lx,
                            lx_stride, D, n, contrib, contrib_stride, stats, stats_next, rhs, zv, round_cnt,
                            slot_handoff, F);
}

// ---------------------------------------------------------------------------
// Forward substitution L y = P b followed by z = D⁻¹ y.
// LDS: items[n_items] 8 B | ptr[n_col+1] u32 | lvl[n_lvl+1] u32 | colperm[n_col] u32 |
//      fcptr[n_col+1] u32 | colsn[n_col] u32 | sn[n_sn] 12 B | sn level ptr u32 |
//      vals[n_items] f64 | y[n_col+1] f64
// Rows of one supernode share a level: pass A takes the items of columns outside the chain
// (all but the LAST pos items of a row), then each chain is finished by one wave
// (chain_solve_wave).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ldlt_fwd_kernel(
    LdltDev L, uint32_t task_base, const double* __restrict__ rhs, int n,
    const double* __restrict__ Lx, long long lx_stride, const double* __restrict__ D,
    double* __restrict__ scontrib, int scontrib_stride, double* __restrict__ zv,
    unsigned int* __restrict__ round_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // round_cnt != nullptr: EVERY round in this launch (one problem; the grid is all tasks in round order, so
  // whatever a workgroup waits for was dispatched before it) — a task waits for the round below it after
  // it has staged its plan and gathered its part of L, and the partial sums cross workgroups coherently
  const bool single = round_cnt != nullptr;
  const uint32_t task_index = task_base + blockIdx.x;
  const LdltTask t = L.tasks[task_index];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  rhs += static_cast<size_t>(b) * n;
  Lx += static_cast<size_t>(b) * lx_stride;
  D += static_cast<size_t>(b) * n;
  zv += static_cast<size_t>(b) * n;
  scontrib += static_cast<size_t>(b) * scontrib_stride;

  SLPX_LDLT_CLOCK(8);
  const uint32_t n_items = t.n_fwd_items;
  const uint32_t g_items = q16(n_items, 2), g_ptr = q16(t.n_col + 1, 4), g_lvl = q16(t.n_lvl + 1, 4),
                 g_cp = q16(t.n_col, 4), g_snd = q16(3 * t.n_sn, 4);
  uint4* s_items = reinterpret_cast<uint4*>(smem_raw);
  uint4* s_ptr = s_items + g_items;
  uint4* s_lvl = s_ptr + g_ptr;
  uint4* s_cp = s_lvl + g_lvl;
  uint4* s_fc = s_cp + g_cp;
  uint4* s_csn = s_fc + g_ptr;
  uint4* s_snd = s_csn + g_cp;
  uint4* s_snl = s_snd + g_snd;
  double* vals = reinterpret_cast<double*>(s_snl + g_lvl);
  double* y = vals + n_items;
  const uint32_t* colsn = reinterpret_cast<const uint32_t*>(s_csn);
  const LdltSn* snd = reinterpret_cast<const LdltSn*>(s_snd);
  const uint32_t* snl = reinterpret_cast<const uint32_t*>(s_snl);
  const uint2* items = reinterpret_cast<const uint2*>(s_items);  // x = lpos, y = ref
  const uint32_t* ptr = reinterpret_cast<const uint32_t*>(s_ptr);
  const uint32_t* lvl = reinterpret_cast<const uint32_t*>(s_lvl);
  const uint32_t* colperm = reinterpret_cast<const uint32_t*>(s_cp);
  const uint32_t* fcptr = reinterpret_cast<const uint32_t*>(s_fc);

  stage16<256>(s_items, reinterpret_cast<const uint4*>(L.fwd_items + t.fwd_item_off), g_items, tid);
  stage16<256>(s_ptr, reinterpret_cast<const uint4*>(L.fwd_ptr + t.colptr_off), g_ptr, tid);
  stage16<256>(s_lvl, reinterpret_cast<const uint4*>(L.col_lvl_ptr + t.lvl_off), g_lvl, tid);
  stage16<256>(s_cp, reinterpret_cast<const uint4*>(L.col_perm + t.col_off), g_cp, tid);
  stage16<256>(s_fc, reinterpret_cast<const uint4*>(L.fwd_contrib_ptr + t.colptr_off), g_ptr, tid);
  stage16<256>(s_csn, reinterpret_cast<const uint4*>(L.col_sn + t.col_off), g_cp, tid);
  stage16<256>(s_snd, reinterpret_cast<const uint4*>(L.sn_desc + t.sn_off), g_snd, tid);
  stage16<256>(s_snl, reinterpret_cast<const uint4*>(L.sn_lvl_ptr + t.lvl_off), g_lvl, tid);
  __syncthreads();
  SLPX_LDLT_CLOCK(9);
  {
    uint32_t q = tid;
    for (; q + 3 * 256 < n_items; q += 4 * 256) {
      const double a0 = Lx[items[q].x], a1 = Lx[items[q + 256].x], a2 = Lx[items[q + 512].x],
                   a3 = Lx[items[q + 768].x];
      vals[q] = a0;
      vals[q + 256] = a1;
      vals[q + 512] = a2;
      vals[q + 768] = a3;
    }
    for (; q < n_items; q += 256) vals[q] = Lx[items[q].x];
  }
  if (single && t.round > 0) round_wait(&round_cnt[t.round - 1], L.round_ptr[t.round] - L.round_ptr[t.round - 1], nullptr);
  {
    const uint32_t* scidx = L.scontrib_idx + t.scontrib_off;
    for (uint32_t i = tid; i < t.n_col; i += 256) {
      double acc = rhs[L.perm[colperm[i]]];
      for (uint32_t c = fcptr[i]; c < fcptr[i + 1]; ++c) acc -= coherent_load(&scontrib[scidx[c]], single);
      y[i] = acc;
    }
  }
  __syncthreads();
  SLPX_LDLT_CLOCK(10);
  // Eight lanes share one row's dot product (a row of L holds ~8 entries on average), so
  // a level costs one LDS round trip per operand instead of one per entry.
  {
    const int lane8 = tid & 7, grp = tid >> 3;
    uint32_t beg = lvl[0], end = t.n_lvl ? lvl[1] : 0;
    for (uint32_t l = 0; l < t.n_lvl; ++l) {
      const uint32_t next_end = lvl[l + 2 <= t.n_lvl ? l + 2 : t.n_lvl];
      for (uint32_t i = beg + grp; i < end; i += 32) {
        const uint32_t qe = ptr[i + 1] - (colsn[i] & 0xffu);  // the chain's own columns come after
        double partial = 0.0;
        for (uint32_t q = ptr[i] + lane8; q < qe; q += 8) partial += vals[q] * y[items[q].y];
        partial = group8_sum(partial);
        if (lane8 == 0) y[i] -= partial;
      }
      __syncthreads();
      const uint32_t sb = snl[l], se = snl[l + 1];
      if (se > sb) {  // chains of two or more columns in this level (block-uniform)
        for (uint32_t q = sb + __builtin_amdgcn_readfirstlane(tid >> 6); q < se; q += 4)
          chain_solve_wave<true>(lds_cast(y), lds_cast(static_cast<const double*>(vals)), lds_cast(ptr), snd[q].col0,
                                 snd[q].w, tid & 63);
        __syncthreads();
      }
      beg = end;
      end = next_end;
    }
  }
  SLPX_LDLT_CLOCK(11);
  // partial sums for rows owned by ancestor tasks
  const uint32_t* sptr = L.sext_ptr + t.sext_ptr_off;
  const LdltSolveItem* sitems = L.sext_items + t.sext_item_off;
  for (uint32_t x = tid >> 3; x < t.n_sext; x += 32) {
    const uint32_t qe = sptr[x + 1];
    double acc = 0.0;
    for (uint32_t q = sptr[x] + (tid & 7); q < qe; q += 8) acc += Lx[sitems[q].lpos] * y[sitems[q].ref];
    acc = group8_sum(acc);
    if ((tid & 7) == 0) coherent_store(&scontrib[L.sext_dst[t.sext_off + x]], acc, single);
  }
  if (single) {
    const int last = L.n_rounds - 1;
    round_signal(round_cnt, static_cast<int>(t.round), L.n_rounds,
                 static_cast<int>(t.round) == last ? L.round_ptr[last + 1] - L.round_ptr[last] : 0u);
  }
  for (uint32_t i = tid; i < t.n_col; i += 256) {
    const uint32_t pj = colperm[i];
    zv[pj] = y[i] / D[pj];
  }
  SLPX_LDLT_CLOCK(12);
}

// ---------------------------------------------------------------------------
// Backward substitution Lᵀ x = z; result un-permuted.  Rows owned by ancestor tasks
// are final (earlier launch): their products are folded into the staged values and
// point at the constant-one slot x[n_col], so the level loop is uniform.
// LDS: items[n_items] 8 B | ptr[n_col+1] u32 | lvl[n_lvl+1] u32 | colperm[n_col] u32 |
//      colsn[n_col] u32 | sn[n_sn] 12 B | sn level ptr u32 | vals[n_items] f64 | x[n_col+1] f64
// A chain's columns share a level: pass A takes the rows below the chain (all but the FIRST
// w - pos - 1 items of a column), then the chain is finished top-down by chain_solve_wave.
// ---------------------------------------------------------------------------
struct BwdCarve {
  uint4 *s_items, *s_rng, *s_lvl, *s_cp, *s_snd, *s_bs;
  uint32_t g_items, g_rng, g_lvl, g_cp, g_snd;
  double *vals, *x;
};
__device__ __forceinline__ BwdCarve bwd_carve(unsigned char* smem, const LdltTask& t) {
  BwdCarve c;
  c.g_items = q16(t.n_bwd_items, 2);
  c.g_rng = q16(t.n_col, 2);
  c.g_lvl = q16(t.n_lvl + 1, 4);
  c.g_cp = q16(t.n_col, 4);
  c.g_snd = q16(3 * t.n_sn, 4);
  c.s_items = reinterpret_cast<uint4*>(smem);
  c.s_rng = c.s_items + c.g_items;
  c.s_lvl = c.s_rng + c.g_rng;
  c.s_cp = c.s_lvl + c.g_lvl;
  c.s_snd = c.s_cp + c.g_cp;
  c.vals = reinterpret_cast<double*>(c.s_snd + c.g_snd);
  c.x = c.vals + t.n_bwd_items;
  // the rows of the back-substitution this task owns (device.hpp: BacksubFuse), behind x[]
  c.s_bs = reinterpret_cast<uint4*>(
      smem + ((static_cast<uint32_t>(reinterpret_cast<unsigned char*>(c.x + t.n_col + 1) - smem) + 15u) & ~15u));
  return c;
}

// the static part of a backward-solve task into LDS (no barrier: the caller's)
template <int THREADS>
__device__ __forceinline__ uint4 ldlt_bwd_stage(const LdltDev& L, const LdltTask& t, uint32_t task_index,
                                                const BwdCarve& c, const BacksubFuse& F) {
  const int tid = threadIdx.x;
  stage16<THREADS>(c.s_items, reinterpret_cast<const uint4*>(L.bwd_items + t.bwd_item_off), c.g_items, tid);
  stage16<THREADS>(c.s_rng, reinterpret_cast<const uint4*>(L.bwd_range + t.col_off), c.g_rng, tid);
  stage16<THREADS>(c.s_lvl, reinterpret_cast<const uint4*>(L.col_lvl_pack + t.lvl_off), c.g_lvl, tid);
  stage16<THREADS>(c.s_cp, reinterpret_cast<const uint4*>(L.col_perm + t.col_off), c.g_cp, tid);
  if (t.n_sn) stage16<THREADS>(c.s_snd, reinterpret_cast<const uint4*>(L.sn_desc + t.sn_off), c.g_snd, tid);
  uint4 bs_task = uint4{0, 0, 0, 0};
  if (F.on) {
    bs_task = F.task_plan[task_index];
    stage16<THREADS>(c.s_bs, F.plan + bs_task.x, bs_task.y, tid);
  }
  return bs_task;
}

// One task of the backward solve, its plan staged (and a barrier passed).
template <int THREADS>
__device__ __forceinline__ void ldlt_bwd_run(const LdltDev& L, const LdltTask& t, uint32_t task_index, int b,
                                             const BwdCarve& c,
                                             const uint4 bs_task, int n, const double* __restrict__ Lx,
                                             long long lx_stride, const double* __restrict__ zv,
                                             double* __restrict__ xg, double* __restrict__ xg_next,
                                             double* __restrict__ out,
                                             unsigned int* __restrict__ round_cnt, const BacksubFuse& F) {
  const int tid = threadIdx.x;
  // xg_next != nullptr (every round in this launch): the ancestors' x is handed over through
  // the values themselves (slot_read) instead of round counters
  const bool by_data = xg_next != nullptr;
  Lx += static_cast<size_t>(b) * lx_stride;
  zv += static_cast<size_t>(b) * n;
  xg += static_cast<size_t>(b) * n;
  if (by_data) xg_next += static_cast<size_t>(b) * n;
  out += static_cast<size_t>(b) * n;
  const uint32_t n_items = t.n_bwd_items;
  double* vals = c.vals;
  double* x = c.x;
  const LdltSn* snd = reinterpret_cast<const LdltSn*>(c.s_snd);
  uint2* items = reinterpret_cast<uint2*>(c.s_items);  // x = lpos, y = ref (rewritten in place)
  const uint2* rng = reinterpret_cast<const uint2*>(c.s_rng);  // {first item below the column's own chain, end}
  const uint32_t* lvl = reinterpret_cast<const uint32_t*>(c.s_lvl);  // column | first chain << 16
  const uint32_t* colperm = reinterpret_cast<const uint32_t*>(c.s_cp);
  SLPX_LDLT_CLOCK(17);
  {
    // L and z are final since the factorization: fetch them BEFORE waiting for the ancestors
    uint32_t q = tid;
    for (; q + 3 * THREADS < n_items; q += 4 * THREADS) {
      const double v0 = Lx[items[q].x], v1 = Lx[items[q + THREADS].x], v2 = Lx[items[q + 2 * THREADS].x],
                   v3 = Lx[items[q + 3 * THREADS].x];
      vals[q] = v0;
      vals[q + THREADS] = v1;
      vals[q + 2 * THREADS] = v2;
      vals[q + 3 * THREADS] = v3;
    }
    for (; q < n_items; q += THREADS) vals[q] = Lx[items[q].x];
    for (uint32_t i = tid; i < t.n_col; i += THREADS) x[i] = zv[colperm[i]];
    if (tid == 0) x[t.n_col] = 1.0;
  }
  // this lane's row of the back-substitution: everything but p is known now
  constexpr int kBsPre = 4;
  const BsRow* bs_rows = reinterpret_cast<const BsRow*>(c.s_bs);
  const BsTerm* bs_terms = reinterpret_cast<const BsTerm*>(c.s_bs + bs_task.w);
  // (only tasks nobody waits for fetch ahead: the others hand x to their descendants first and do
  // their rows afterwards, off the critical path — and must not stall here on these loads)
  const bool bs_mine = static_cast<uint32_t>(tid) < bs_task.z && t.round == 0;
  BsRow bs_row = BsRow{0, 0};
  double bs_a[kBsPre], bs_p[kBsPre], bs_s = 1.0, bs_z = 0.0, bs_ci = 0.0;
  uint32_t bs_ref[kBsPre];
#pragma unroll
  for (int k = 0; k < kBsPre; ++k) {
    bs_a[k] = 0.0;
    bs_p[k] = 0.0;
    bs_ref[k] = 0;
  }
  if (bs_mine) {
    bs_row = bs_rows[tid];
    const uint32_t first = bs_row.terms & 0xfffffu, cnt = bs_row.terms >> 20;
    bs_s = F.s[bs_row.r];
    bs_z = F.z[bs_row.r];
    bs_ci = F.V[F.off_ci + bs_row.r];
#pragma unroll
    for (int k = 0; k < kBsPre; ++k)
      if (static_cast<uint32_t>(k) < cnt) {
        const BsTerm bt = bs_terms[first + k];
        bs_a[k] = F.V[bt.a];
        bs_ref[k] = bt.ref;
      }
  }
  // rows owned by ancestor tasks (later rounds) must be final before they are gathered
  if (!by_data && round_cnt != nullptr && static_cast<int>(t.round) + 1 < L.n_rounds)
    round_wait(&round_cnt[b * L.n_rounds + t.round + 1],
               L.round_ptr[t.round + 2] - L.round_ptr[t.round + 1], nullptr);
  {
    // fold the finished x of ancestor rows into the staged values; those items then point at
    // the constant-one slot (each thread touches the items it staged itself: no barrier needed)
    for (uint32_t q = tid; q < n_items; q += THREADS) {
      const uint32_t ref = items[q].y;
      if (ref & 0x80000000u) {
        const double* src = &xg[ref & 0x7fffffffu];
        vals[q] *= by_data ? slot_read(src) : coherent_load(src, round_cnt != nullptr);
        items[q].y = t.n_col;
      }
    }
    if (bs_mine) {
#pragma unroll
      for (int k = 0; k < kBsPre; ++k)
        if (bs_ref[k] & 0x80000000u) {
          const double* src = &xg[bs_ref[k] & 0x7fffffffu];
          bs_p[k] = by_data ? slot_read(src) : coherent_load(src, round_cnt != nullptr);
        }
    }
  }
  __syncthreads();
  SLPX_LDLT_CLOCK(18);
  // A level holds a handful of columns (3-4 on average; a dozen with supernodal levels).  Tasks
  // without a chain of two or more columns: ONE wave runs the whole level loop — eight 8-lane
  // groups — with no workgroup barrier between two levels (the wave's own LDS operations are
  // issued in order, only the compiler has to be kept from moving them across the level
  // boundary).  Tasks with chains: every wave, one barrier after the gather and one after the
  // chains, which get a wave each (a barrier of four waves is ~40 clocks; solving the chains
  // of a level one after the other in a single wave cost more than the levels it saved:
  // backward solve at N=5000 39.9 -> 52.5 us).
  if (t.n_sn == 0) {
    if (tid < 64) {
      const int lane8 = tid & 7, grp = tid >> 3;
      uint32_t end = lvl[t.n_lvl] & 0xffffu, beg = t.n_lvl ? lvl[t.n_lvl - 1] & 0xffffu : 0;
      for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
        const uint32_t next_beg = lvl[l >= 1 ? l - 1 : 0] & 0xffffu;
        for (uint32_t i = beg + grp; i < end; i += 8) {
          const uint2 r = rng[i];
          double partial = 0.0;
          for (uint32_t q = r.x + lane8; q < r.y; q += 8) partial += vals[q] * x[items[q].y];
          partial = group8_sum(partial);
          if (lane8 == 0) x[i] -= partial;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        end = beg;
        beg = next_beg;
      }
    }
  } else {
    const int lane8 = tid & 7, grp = tid >> 3;
    uint32_t end_pack = lvl[t.n_lvl], beg_pack = t.n_lvl ? lvl[t.n_lvl - 1] : 0;
    for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
      const uint32_t next_pack = lvl[l >= 1 ? l - 1 : 0];
      const uint32_t beg = beg_pack & 0xffffu, end = end_pack & 0xffffu;
      for (uint32_t i = beg + grp; i < end; i += THREADS / 8) {
        const uint2 r = rng[i];
        double partial = 0.0;
        for (uint32_t q = r.x + lane8; q < r.y; q += 8) partial += vals[q] * x[items[q].y];
        partial = group8_sum(partial);
        if (lane8 == 0) x[i] -= partial;
      }
      __syncthreads();
      const uint32_t sb = beg_pack >> 16, se = end_pack >> 16;
      if (se > sb) {  // chains of two or more columns in this level (block-uniform)
        for (uint32_t q = sb + __builtin_amdgcn_readfirstlane(tid >> 6); q < se; q += THREADS / 64)
          chain_solve_wave<false>(lds_cast(x), lds_cast(static_cast<const double*>(vals)), lds_cast(rng), snd[q].col0,
                                  snd[q].w, tid & 63);
        __syncthreads();
      }
      end_pack = beg_pack;
      beg_pack = next_pack;
    }
  }
  __syncthreads();
  SLPX_LDLT_CLOCK(19);
  // the descendants wait for x alone: hand it over before anything else goes out
  for (uint32_t i = tid; i < t.n_col; i += THREADS)
    coherent_store(&xg[colperm[i]], x[i], round_cnt != nullptr || by_data);
  if (by_data) {
    for (uint32_t i = tid; i < t.n_col; i += THREADS)
      coherent_store(&xg_next[colperm[i]], __longlong_as_double(static_cast<long long>(kSlotEmpty)), true);
  } else if (round_cnt != nullptr) {
    round_signal(&round_cnt[b * L.n_rounds], static_cast<int>(t.round), L.n_rounds,
                 t.round == 0 ? L.round_ptr[1] - L.round_ptr[0] : 0u);
  }
  for (uint32_t i = tid; i < t.n_col; i += THREADS) out[L.perm[colperm[i]]] = x[i];
  if (F.on) {
    const double m = F.mu[0];
    auto p_of = [&](uint32_t ref) {
      return (ref & 0x80000000u) ? coherent_load(&xg[ref & 0x7fffffffu], round_cnt != nullptr || by_data) : x[ref];
    };
    for (uint32_t j = tid; j < bs_task.z; j += THREADS) {
      const bool ahead = bs_mine && j == static_cast<uint32_t>(tid);
      const BsRow row = ahead ? bs_row : bs_rows[j];
      const uint32_t first = row.terms & 0xfffffu, cnt = row.terms >> 20;
      double aipx = 0.0, s_r = bs_s, z_r = bs_z, ci_r = bs_ci;
      uint32_t k0 = 0;
      if (ahead) {
#pragma unroll
        for (int k = 0; k < kBsPre; ++k)
          if (static_cast<uint32_t>(k) < cnt)
            aipx = backsub_dot(aipx, bs_a[k], (bs_ref[k] & 0x80000000u) ? bs_p[k] : x[bs_ref[k]]);
        k0 = kBsPre;
      } else {  // (nothing was fetched ahead)
        s_r = F.s[row.r];
        z_r = F.z[row.r];
        ci_r = F.V[F.off_ci + row.r];
      }
      for (uint32_t k = k0; k < cnt; ++k) {
        const BsTerm bt = bs_terms[first + k];
        aipx = backsub_dot(aipx, F.V[bt.a], p_of(bt.ref));
      }
      backsub_row(ci_r, s_r, z_r, m, aipx, &F.ps[row.r], &F.pz[row.r]);
    }
  }
  SLPX_LDLT_CLOCK(20);
}

// the factorization's inertia counters to the host (pinned memory) with a sequence number it spins on
__device__ __forceinline__ void publish_stats(const BacksubFuse& F, bool coherent) {
  LdltStats st;
  if (coherent) {
    const int* src = reinterpret_cast<const int*>(F.stats_src);
    st.n_pos = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st.n_neg = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st.n_zero = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st.n_bad = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st.min_abs_bits = __hip_atomic_load(&F.stats_src->min_abs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    st = F.stats_src[0];
  }
  F.stats_host[0] = st;
  if (F.seq_host != nullptr) {
    __threadfence_system();
    const unsigned long long v = *F.seq_dev + 1;
    *F.seq_dev = v;
    *F.seq_host = v;
  }
}

// a second attempt's counters beside them (ldlt_mf_twin_kernel); before publish_stats, whose fence and sequence
// number cover both
__device__ __forceinline__ void publish_stats_copy(const LdltStats* stats, LdltStats* host) {
  LdltStats st;
  const int* src = reinterpret_cast<const int*>(stats);
  st.n_pos = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  st.n_neg = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  st.n_zero = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  st.n_bad = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  st.min_abs_bits = __hip_atomic_load(&stats->min_abs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  host[0] = st;
}

__global__ __launch_bounds__(256) void ldlt_bwd_kernel(
    LdltDev L, uint32_t task_base, int n, const double* __restrict__ Lx, long long lx_stride,
    const double* __restrict__ zv, double* __restrict__ xg, double* __restrict__ xg_next,
    double* __restrict__ out, unsigned int* __restrict__ round_cnt, BacksubFuse F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // the verdict of the factorization this solve belongs to: known since the launch began; handed
  // over by the last workgroup, a leaf task that has to wait for its ancestors anyway
  if (F.on && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && F.stats_host != nullptr) publish_stats(F, false);
  // single-launch mode (round_cnt != nullptr): every task of every round, LAST round first
  const uint32_t task_index =
      (round_cnt != nullptr || xg_next != nullptr) ? task_base - blockIdx.x : task_base + blockIdx.x;
  const LdltTask t = L.tasks[task_index];
  SLPX_LDLT_CLOCK(16);
  const BwdCarve c = bwd_carve(smem_raw, t);
  const uint4 bs_task = ldlt_bwd_stage<256>(L, t, task_index, c, F);
  __syncthreads();
  ldlt_bwd_run<256>(L, t, task_index, blockIdx.y, c, bs_task, n, Lx, lx_stride, zv, xg, xg_next, out, round_cnt, F);
}

}  // namespace slpx
